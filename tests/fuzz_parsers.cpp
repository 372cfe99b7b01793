// Mutation test of the host layer's file parsers (PNG, JPEG, Radiance .hdr, glTF): every file named on the command
// line is corrupted a few bytes at a time (random bytes, bit flips, 00 / FF, truncation) and parsed again; a parser
// must either produce an image of the size it reports or throw.  The first pass over a file parses it unchanged.  Built with -fsanitize=address,undefined by
// tests/test_host_layer.py, so that an out-of-bounds access or an overflow ends the run.
// usage: fuzz_parsers <scratch dir> <files...>     (ITERS and SEED in the environment)
#include "host.h"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
using namespace ptamd;
static std::vector<uint8_t> rd(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), {}); }
int main(int argc, char** argv)
{
    std::mt19937 rng(getenv("SEED") ? atoi(getenv("SEED")) : 12345);
    const int base_iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 1000;
    if (argc < 3) return 1;
    const std::string scratch = argv[1];
    long ok = 0, bad = 0;
    for (int a = 2; a < argc; a++) {
        std::vector<uint8_t> src = rd(argv[a]);
        std::string name = argv[a];
        bool hdr = name.size() > 4 && name.substr(name.size() - 4) == ".hdr";
        bool gltf = name.size() > 5 && name.substr(name.size() - 5) == ".gltf";
        if (src.empty()) { printf("cannot read %s\n", argv[a]); return 1; }
        int iters = (hdr || gltf) ? base_iters / 10 + 1 : base_iters;
        for (int it = 0; it < iters; it++) {
            std::vector<uint8_t> b = src;
            int nm = it == 0 ? 0 : 1 + rng() % 8;       // the first pass parses the file as it is (hand-made hostile files)
            for (int k = 0; k < nm; k++) {
                size_t pos = rng() % b.size();
                switch (rng() % 4) {
                case 0: b[pos] = (uint8_t)rng(); break;
                case 1: b[pos] ^= (uint8_t)(1u << (rng() % 8)); break;
                case 2: b[pos] = (rng() & 1) ? 0xFF : 0x00; break;
                default: if (b.size() > 16) b.resize(pos + 1); break;     // truncate
                }
                if (b.empty()) break;
            }
            try {
                if (hdr || gltf) {
                    std::string tmp = scratch + "/in" + (hdr ? ".hdr" : ".gltf");
                    { std::ofstream o(tmp, std::ios::binary); o.write((const char*)b.data(), (std::streamsize)b.size()); }
                    if (hdr) { HdrImage h = LoadHdr(tmp); (void)h; }
                    else { MeshSource m(tmp); (void)m; }
                } else {
                    TextureData t = DecodeImage(b);
                    if (t.RGBA8.size() != (size_t)t.Width * t.Height * 4) { printf("size mismatch\n"); return 2; }
                }
                ok++;
            } catch (const std::exception&) { bad++; }
        }
    }
    printf("parsed %ld mutated inputs, rejected %ld\n", ok, bad);
    return 0;
}
namespace ptamd { Mat4 Mat4::Identity() { Mat4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; } }
