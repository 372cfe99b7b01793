// Issue cost of wave64 VALU instructions on gfx950: cycles per instruction per SIMD with 4 and 8 resident waves per SIMD
// (independent instructions, no memory traffic; the clock is taken as 2.4 GHz after a warm-up).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define KERNEL(NAME, ASM)                                                                                   \
    __global__ void __launch_bounds__(256) NAME(float* out, int iters, float a, float b)                   \
    {                                                                                                       \
        float x[16];                                                                                        \
        typedef float f2 __attribute__((ext_vector_type(2)));                                               \
        f2 y[16]; f2 c2 = {a, b};                                                                           \
        for (int i = 0; i < 16; i++) { x[i] = threadIdx.x * 0.001f + i; y[i] = f2{x[i], 1.0f}; }             \
        for (int it = 0; it < iters; it++) {                                                                \
            _Pragma("unroll") for (int i = 0; i < 16; i++) { ASM; }                                         \
        }                                                                                                   \
        float s = 0;                                                                                        \
        for (int i = 0; i < 16; i++) s += x[i] + y[i][0] + y[i][1];                                         \
        if (s == 12345.678f) out[0] = s + c2[0];                                                            \
    }
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)))
KERNEL(k_pkfma, asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(c2)))
KERNEL(k_cvt, asm volatile("v_cvt_f32_ubyte1_e32 %0, %0" : "+v"(x[i])))
KERNEL(k_fmamix, asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(x[i]) : "v"(a), "v"(b)))
KERNEL(k_perm, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)))
KERNEL(k_max3, asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)))
KERNEL(k_maxu, asm volatile("v_max_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a)))
KERNEL(k_maxf, asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)))
KERNEL(k_andor, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)))
KERNEL(k_mul, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)))
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(0x5555555555555555ull)))
KERNEL(k_cmp, asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc"))
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i])))
KERNEL(k_mullo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a)))
KERNEL(k_lshladd, asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x[i]) : "v"(a)))
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(a)))
KERNEL(k_add, asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)))
KERNEL(k_cvtf16, asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(x[i])))
typedef void (*kern_t)(float*, int, float, float);
void run(const char* name, kern_t k)
{
    float* d; hipMalloc(&d, 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int iters = 100000;
    for (int w = 0; w < 10; w++) k<<<p.multiProcessorCount * 4, 256>>>(d, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    printf("%-12s", name);
    for (int wps : {2, 4, 8}) {
        int blocks = p.multiProcessorCount * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  %d waves/SIMD: %5.2f cyc", wps, ms * 1e-3 * 2.4e9 / ((double)wps * iters * 16));
    }
    printf("\n");
    hipFree(d);
}
int main()
{
#define R(n) run(#n, n)
    R(k_fma); R(k_pkfma); R(k_cvt); R(k_fmamix); R(k_perm); R(k_max3); R(k_maxu); R(k_maxf); R(k_andor); R(k_mul); R(k_cndmask); R(k_cmp);
    R(k_rcp); R(k_mullo); R(k_lshladd); R(k_mov); R(k_add); R(k_cvtf16);
    return 0;
}
