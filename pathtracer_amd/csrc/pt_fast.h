// Stated-tolerance arithmetic for the shading stage (pt_set_precision(ctx, PT_PRECISION_FAST)).
//
// The default kernels evaluate every floating-point result with correctly rounded IEEE operations in one fixed order, so
// that frames equal the CPU oracle's bit for bit; the price is that a division costs about ten vector instructions, a
// square root about twenty, and sin / cos / log / exp are polynomial sequences.  GLSL itself promises far less for the
// reference's shaders (division 2.5 ulp, inversesqrt 2 ulp, sin / cos an absolute 2^-11, pow / log / exp a few ulp), so this
// namespace compiles the same shading text (pt_bsdf.inc, pt_shade.inc) a second time with the hardware's single-instruction
// operations: v_rcp_f32, v_sqrt_f32, v_rsq_f32 (1 ulp), v_sin_f32 / v_cos_f32, v_log_f32 / v_exp_f32, and with
// multiply-adds fused.  Camera rays, the traversal and the accumulation stay bit-exact, so primary hits and the geometry
// of every individual ray query are unchanged; what changes is the last bits of the BSDF values and sampled directions, i.e.
// per-sample radiance within the tolerance the tests state (tests/test_gpu_precision.py) and a converged image within
// relRMSE 1e-3 of the bit-exact one.
//
// Included by trace.hip inside namespace ptd, after the bit-exact copy of the same text.
namespace fast {

#pragma clang fp contract(fast)

PT_DEV float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
PT_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// v_sin_f32 / v_cos_f32 take revolutions; arguments here are angles in [0, 2 pi]
PT_DEV void det_sincos(float x, float& s_out, float& c_out)
{
    const float r = x * 0.15915494309189535f;
    s_out = __builtin_amdgcn_sinf(r);
    c_out = __builtin_amdgcn_cosf(r);
}
PT_DEV float det_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }         // v_log_f32 is log2
PT_DEV float det_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
PT_DEV float det_pow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }

#define PT_INVERSESQRT(x) __builtin_amdgcn_rsqf(x)
#include "pt_vec.inc"
#undef PT_INVERSESQRT

#include "pt_shade.inc"

#pragma clang fp contract(off)

} // namespace fast
