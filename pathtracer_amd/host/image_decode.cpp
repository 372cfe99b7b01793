// Texture decoders of the host layer: PNG (all colour types, 8 and 16 bit, no interlace) and baseline JPEG
// (SOF0, Huffman, 8-bit, 1 or 3 components, any sampling factors with replicated chroma).  The reference gets its
// texels from the absent engine's image loader (glTF images of MeshSource, bound at RayTracingLayer.cpp:262-263);
// its assets use exactly these two formats (Suzanne: PNG; Sponza: 65 baseline 4:4:4 JPEGs and 4 PNGs).
//
// PNG follows the PNG specification (chunks, zlib stream through zlib's inflate, the five scanline filters, palette
// and tRNS); 16-bit samples keep their high byte.  JPEG follows ITU-T T.81: marker parsing, Huffman decoding,
// dequantisation, the separable 8x8 inverse DCT of Loeffler, Ligtenberg and Moschytz in 13-bit fixed point with the
// usual two-pass scaling, and the JFIF YCbCr -> RGB conversion in 16-bit fixed point.
#include "host.h"
#include <cstring>
#include <stdexcept>
#include <zlib.h>

namespace ptamd {

namespace {

std::vector<uint8_t> read_file(const std::string& path)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> d;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    return d;
}

const size_t MAX_PIXELS = (size_t)1 << 26;     // a header claiming more is refused before anything is allocated

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// ---- PNG ----------------------------------------------------------------------------------------------------
int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

TextureData decode_png(const std::vector<uint8_t>& d)
{
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (d.size() < 8 || memcmp(d.data(), sig, 8)) throw std::runtime_error("png: bad signature");
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool have_ihdr = false, done = false;
    size_t pos = 8;
    while (!done) {
        if (pos + 12 > d.size()) throw std::runtime_error("png: truncated");
        uint32_t len = be32(&d[pos]);
        if (pos + 12 + (size_t)len > d.size()) throw std::runtime_error("png: chunk exceeds the file");
        const uint8_t* type = &d[pos + 4];
        const uint8_t* body = &d[pos + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) throw std::runtime_error("png: bad IHDR");
            w = be32(body); h = be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] || body[11]) throw std::runtime_error("png: unknown compression or filter method");
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "IEND", 4)) done = true;
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || w == 0 || h == 0 || (uint64_t)w * h > MAX_PIXELS) throw std::runtime_error("png: bad size");
    if (interlace) throw std::runtime_error("png: interlaced images are not supported");
    int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                    (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                    ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!channels || !depth_ok) throw std::runtime_error("png: unsupported colour type / bit depth");
    if (ctype == 3 && plte.size() < 3) throw std::runtime_error("png: palette missing");
    size_t bpp_bits = (size_t)channels * depth, stride = ((size_t)w * bpp_bits + 7) / 8, bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf out_len = (uLongf)raw.size();
    int zr = uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || out_len != raw.size()) throw std::runtime_error("png: image data does not inflate to the image size");
    // undo the scanline filters in place
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < h; y++) {
        uint8_t* row = &raw[(stride + 1) * y + 1];
        const uint8_t* up = y ? &raw[(stride + 1) * (y - 1) + 1] : zero.data();
        int filter = raw[(stride + 1) * y];
        for (size_t x = 0; x < stride; x++) {
            int a = x >= bpp ? row[x - bpp] : 0, b = up[x], c = x >= bpp ? up[x - bpp] : 0;
            int v = row[x];
            switch (filter) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: throw std::runtime_error("png: unknown filter type");
            }
            row[x] = (uint8_t)v;
        }
    }
    TextureData t;
    t.Width = w; t.Height = h;
    t.RGBA8.resize((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* row = &raw[(stride + 1) * y + 1];
        uint8_t* out = &t.RGBA8[(size_t)y * w * 4];
        auto sample = [&](size_t x, int ch) -> uint32_t {      // raw sample value, full bit depth
            if (depth == 8) return row[x * channels + ch];
            if (depth == 16) return ((uint32_t)row[(x * channels + ch) * 2] << 8) | row[(x * channels + ch) * 2 + 1];
            size_t bit = x * depth;
            return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
        };
        auto to8 = [&](uint32_t v) -> uint8_t {
            if (depth == 8) return (uint8_t)v;
            if (depth == 16) return (uint8_t)(v >> 8);
            return (uint8_t)(v * 255u / ((1u << depth) - 1u));
        };
        for (uint32_t x = 0; x < w; x++) {
            uint8_t r, g, b, a = 255;
            if (ctype == 3) {
                uint32_t i = sample(x, 0);
                if ((size_t)i * 3 + 2 >= plte.size()) throw std::runtime_error("png: palette index out of range");
                r = plte[i * 3]; g = plte[i * 3 + 1]; b = plte[i * 3 + 2];
                if (i < trns.size()) a = trns[i];
            } else if (ctype == 0 || ctype == 4) {
                uint32_t v = sample(x, 0);
                r = g = b = to8(v);
                if (ctype == 4) a = to8(sample(x, 1));
                else if (trns.size() >= 2 && v == (((uint32_t)trns[0] << 8) | trns[1])) a = 0;
            } else {
                uint32_t vr = sample(x, 0), vg = sample(x, 1), vb = sample(x, 2);
                r = to8(vr); g = to8(vg); b = to8(vb);
                if (ctype == 6) a = to8(sample(x, 3));
                else if (trns.size() >= 6 && vr == (((uint32_t)trns[0] << 8) | trns[1]) && vg == (((uint32_t)trns[2] << 8) | trns[3]) &&
                         vb == (((uint32_t)trns[4] << 8) | trns[5])) a = 0;
            }
            out[x * 4] = r; out[x * 4 + 1] = g; out[x * 4 + 2] = b; out[x * 4 + 3] = a;
        }
    }
    return t;
}

// ---- baseline JPEG --------------------------------------------------------------------------------------------
struct Huff {
    // canonical code tables of T.81 Annex C: for code length l, codes [mincode[l], maxcode[l]] map to symbols from valptr[l]
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
};

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint32_t acc = 0; int nbits = 0;
    bool marker_hit = false;
    int bit()
    {
        if (nbits == 0) {
            uint8_t byte = 0;
            if (!marker_hit && p < end) {
                byte = *p++;
                if (byte == 0xFF) {
                    if (p < end && *p == 0x00) p++;                 // stuffed zero
                    else { marker_hit = true; p--; byte = 0; }      // a marker: feed zeros from here on
                }
            }
            acc = byte; nbits = 8;
        }
        nbits--;
        return (acc >> nbits) & 1;
    }
    int bits(int n) { int v = 0; for (int i = 0; i < n; i++) v = (v << 1) | bit(); return v; }
    void reset() { acc = 0; nbits = 0; marker_hit = false; }
};

int decode_symbol(BitReader& br, const Huff& h)
{
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    throw std::runtime_error("jpeg: bad Huffman code");
}

int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }

// a dequantised coefficient; corrupt streams can drive the DC predictor anywhere, so the product is formed in 64 bits
// and kept inside the range the inverse transform's 64-bit intermediates are safe for
int dequant(long v, int q)
{
    long x = v * q;
    const long LIM = 1L << 24;
    return (int)(x < -LIM ? -LIM : (x > LIM ? LIM : x));
}

const int ZIGZAG[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// 8x8 inverse DCT, Loeffler-Ligtenberg-Moschytz factorisation, 13-bit constants, 2 extra bits kept between the passes
void idct8x8(const int* in /* dequantised, natural order */, uint8_t* out, int out_stride)
{
    const int CONST_BITS = 13, PASS1_BITS = 2;
    const int F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633, F_1_501 = 12299,
              F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
    int ws[64];
    auto descale = [](long x, int n) { return (int)((x + (1l << (n - 1))) >> n); };
    for (int c = 0; c < 8; c++) {                       // columns
        const int* p = in + c;
        int* w = ws + c;
        if (!(p[8] | p[16] | p[24] | p[32] | p[40] | p[48] | p[56])) {
            int dc = p[0] * (1 << PASS1_BITS);
            for (int r = 0; r < 8; r++) w[r * 8] = dc;
            continue;
        }
        long z2 = p[16], z3 = p[48];
        long z1 = (z2 + z3) * F_0_541;
        long tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
        z2 = p[0]; z3 = p[32];
        long tmp0 = (z2 + z3) * (1L << CONST_BITS), tmp1 = (z2 - z3) * (1L << CONST_BITS);
        long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = p[56]; tmp1 = p[40]; tmp2 = p[24]; tmp3 = p[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3, z5 = (z3 + z4) * F_1_175;
        tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        w[0] = descale(tmp10 + tmp3, CONST_BITS - PASS1_BITS); w[56] = descale(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
        w[8] = descale(tmp11 + tmp2, CONST_BITS - PASS1_BITS); w[48] = descale(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
        w[16] = descale(tmp12 + tmp1, CONST_BITS - PASS1_BITS); w[40] = descale(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
        w[24] = descale(tmp13 + tmp0, CONST_BITS - PASS1_BITS); w[32] = descale(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
    }
    auto clamp8 = [](int v) { v += 128; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    for (int r = 0; r < 8; r++) {                       // rows
        const int* w = ws + r * 8;
        uint8_t* o = out + r * out_stride;
        long z2 = w[2], z3 = w[6];
        long z1 = (z2 + z3) * F_0_541;
        long tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
        long tmp0 = ((long)w[0] + w[4]) * (1L << CONST_BITS), tmp1 = ((long)w[0] - w[4]) * (1L << CONST_BITS);
        long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3, z5 = (z3 + z4) * F_1_175;
        tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        const int S = CONST_BITS + PASS1_BITS + 3;
        o[0] = clamp8(descale(tmp10 + tmp3, S)); o[7] = clamp8(descale(tmp10 - tmp3, S));
        o[1] = clamp8(descale(tmp11 + tmp2, S)); o[6] = clamp8(descale(tmp11 - tmp2, S));
        o[2] = clamp8(descale(tmp12 + tmp1, S)); o[5] = clamp8(descale(tmp12 - tmp1, S));
        o[3] = clamp8(descale(tmp13 + tmp0, S)); o[4] = clamp8(descale(tmp13 - tmp0, S));
    }
}

TextureData decode_jpeg(const std::vector<uint8_t>& d)
{
    if (d.size() < 4 || d[0] != 0xFF || d[1] != 0xD8) throw std::runtime_error("jpeg: no SOI marker");
    uint16_t qt[4][64] = {};
    bool have_qt[4] = {};
    Huff hdc[4], hac[4];
    struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0, bw = 0, bh = 0; std::vector<uint8_t> plane; };
    std::vector<Comp> comp;
    int width = 0, height = 0, restart_interval = 0;
    bool have_sof = false;
    size_t pos = 2;
    TextureData t;
    for (;;) {
        if (pos + 4 > d.size()) throw std::runtime_error("jpeg: truncated before the scan");
        if (d[pos] != 0xFF) throw std::runtime_error("jpeg: marker expected");
        while (pos + 2 < d.size() && d[pos + 1] == 0xFF) pos++;      // fill bytes
        uint8_t m = d[pos + 1];
        pos += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) throw std::runtime_error("jpeg: no scan before EOI");
        if (pos + 2 > d.size()) throw std::runtime_error("jpeg: truncated segment");
        size_t len = ((size_t)d[pos] << 8) | d[pos + 1];
        if (len < 2 || pos + len > d.size()) throw std::runtime_error("jpeg: segment exceeds the file");
        const uint8_t* s = &d[pos + 2];
        size_t n = len - 2;
        if (m == 0xDB) {                                             // quantisation tables
            size_t k = 0;
            while (k < n) {
                int pq = s[k] >> 4, tq = s[k] & 15;
                k++;
                if (tq > 3 || k + (pq ? 128 : 64) > n) throw std::runtime_error("jpeg: bad DQT");
                for (int i = 0; i < 64; i++) { qt[tq][ZIGZAG[i]] = pq ? (uint16_t)((s[k] << 8) | s[k + 1]) : s[k]; k += pq ? 2 : 1; }
                have_qt[tq] = true;
            }
        } else if (m == 0xC4) {                                      // Huffman tables
            size_t k = 0;
            while (k < n) {
                if (k + 17 > n) throw std::runtime_error("jpeg: bad DHT");
                int tc = s[k] >> 4, th = s[k] & 15;
                if (tc > 1 || th > 3) throw std::runtime_error("jpeg: bad DHT class or id");
                Huff& h = tc ? hac[th] : hdc[th];
                int counts[17] = {0}, total = 0;
                for (int l = 1; l <= 16; l++) { counts[l] = s[k + l]; total += counts[l]; }
                k += 17;
                if (total > 256 || k + total > n) throw std::runtime_error("jpeg: bad DHT length");
                memcpy(h.vals, s + k, (size_t)total);
                k += (size_t)total;
                int code = 0, idx = 0;
                for (int l = 1; l <= 16; l++) {
                    h.valptr[l] = idx; h.mincode[l] = code;
                    code += counts[l]; idx += counts[l];
                    h.maxcode[l] = counts[l] ? code - 1 : -1;
                    code <<= 1;
                }
                h.present = true;
            }
        } else if (m == 0xC0 || m == 0xC1) {                         // baseline / extended sequential, Huffman
            if (n < 6 || s[0] != 8) throw std::runtime_error("jpeg: only 8-bit samples are supported");
            height = (s[1] << 8) | s[2]; width = (s[3] << 8) | s[4];
            int nc = s[5];
            if ((nc != 1 && nc != 3) || n < 6 + 3 * (size_t)nc || width == 0 || height == 0) throw std::runtime_error("jpeg: unsupported component count or size");
            if ((size_t)width * (size_t)height > MAX_PIXELS) throw std::runtime_error("jpeg: image larger than 2^26 pixels");
            comp.resize((size_t)nc);
            for (int i = 0; i < nc; i++) {
                comp[i].id = s[6 + 3 * i]; comp[i].h = s[7 + 3 * i] >> 4; comp[i].v = s[7 + 3 * i] & 15; comp[i].tq = s[8 + 3 * i];
                if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) throw std::runtime_error("jpeg: bad sampling factors");
            }
            have_sof = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            throw std::runtime_error("jpeg: only baseline (SOF0/SOF1) images are supported");
        } else if (m == 0xDD) {
            if (n < 2) throw std::runtime_error("jpeg: bad DRI");
            restart_interval = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {                                      // start of scan: decode the whole image
            if (!have_sof) throw std::runtime_error("jpeg: scan before frame header");
            int ns = s[0];
            if (ns != (int)comp.size() || n < 1 + 2 * (size_t)ns + 3) throw std::runtime_error("jpeg: only single-scan images are supported");
            for (int i = 0; i < ns; i++) {
                int cid = s[1 + 2 * i];
                Comp* c = nullptr;
                for (Comp& cc : comp) if (cc.id == cid) c = &cc;
                if (!c) throw std::runtime_error("jpeg: scan names an unknown component");
                c->td = s[2 + 2 * i] >> 4; c->ta = s[2 + 2 * i] & 15;
                if (c->td > 3 || c->ta > 3 || !hdc[c->td].present || !hac[c->ta].present || !have_qt[c->tq]) throw std::runtime_error("jpeg: scan uses an undefined table");
            }
            int hmax = 1, vmax = 1;
            for (Comp& c : comp) { if (c.h > hmax) hmax = c.h; if (c.v > vmax) vmax = c.v; }
            int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
            for (Comp& c : comp) { c.bw = mcux * c.h * 8; c.bh = mcuy * c.v * 8; c.plane.assign((size_t)c.bw * c.bh, 0); c.pred = 0; }
            BitReader br{&d[pos + len], d.data() + d.size()};
            int until_restart = restart_interval;
            for (int my = 0; my < mcuy; my++)
                for (int mx = 0; mx < mcux; mx++) {
                    if (restart_interval && until_restart == 0) {
                        // byte-align, expect RSTn
                        br.reset();
                        const uint8_t* q = br.p;
                        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
                        if (q + 1 >= br.end) throw std::runtime_error("jpeg: restart marker missing");
                        br.p = q + 2;
                        for (Comp& c : comp) c.pred = 0;
                        until_restart = restart_interval;
                    }
                    for (Comp& c : comp)
                        for (int by = 0; by < c.v; by++)
                            for (int bx = 0; bx < c.h; bx++) {
                                int blk[64] = {0};
                                int tdc = decode_symbol(br, hdc[c.td]);
                                if (tdc > 11) throw std::runtime_error("jpeg: bad DC size");
                                c.pred += extend(br.bits(tdc), tdc);
                                if (c.pred > (1 << 20)) c.pred = 1 << 20;
                                if (c.pred < -(1 << 20)) c.pred = -(1 << 20);
                                blk[0] = dequant(c.pred, qt[c.tq][0]);
                                for (int k = 1; k < 64;) {
                                    int rs = decode_symbol(br, hac[c.ta]);
                                    int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) {
                                        if (r == 15) { k += 16; continue; }
                                        break;                          // end of block
                                    }
                                    k += r;
                                    if (k > 63) throw std::runtime_error("jpeg: coefficient index out of range");
                                    blk[ZIGZAG[k]] = dequant(extend(br.bits(sz), sz), qt[c.tq][ZIGZAG[k]]);
                                    k++;
                                }
                                idct8x8(blk, &c.plane[(size_t)((my * c.v + by) * 8) * c.bw + (size_t)(mx * c.h + bx) * 8], c.bw);
                            }
                    if (restart_interval) until_restart--;
                }
            // colour conversion (JFIF): chroma of subsampled components is replicated
            t.Width = (uint32_t)width; t.Height = (uint32_t)height;
            t.RGBA8.resize((size_t)width * height * 4);
            const int SCALE = 16, HALF = 1 << (SCALE - 1);
            auto fix = [](double x) { return (int)(x * 65536.0 + 0.5); };
            const int CR_R = fix(1.40200), CB_B = fix(1.77200), CR_G = -fix(0.71414), CB_G = -fix(0.34414);
            auto clamp8 = [](int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
            for (int y = 0; y < height; y++)
                for (int x = 0; x < width; x++) {
                    auto at = [&](const Comp& c) { return (int)c.plane[(size_t)(y * c.v / vmax) * c.bw + (size_t)(x * c.h / hmax)]; };
                    uint8_t* o = &t.RGBA8[((size_t)y * width + x) * 4];
                    int Y = at(comp[0]);
                    if (comp.size() == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; }
                    else {
                        int cb = at(comp[1]) - 128, cr = at(comp[2]) - 128;
                        o[0] = clamp8(Y + ((CR_R * cr + HALF) >> SCALE));
                        o[1] = clamp8(Y + ((CB_G * cb + CR_G * cr + HALF) >> SCALE));
                        o[2] = clamp8(Y + ((CB_B * cb + HALF) >> SCALE));
                    }
                    o[3] = 255;
                }
            return t;
        }
        pos += len;
    }
}

} // namespace

TextureData DecodeImage(const std::vector<uint8_t>& bytes)
{
    if (bytes.size() >= 8 && bytes[0] == 137 && bytes[1] == 'P') return decode_png(bytes);
    if (bytes.size() >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8) return decode_jpeg(bytes);
    throw std::runtime_error("image: neither PNG nor JPEG");
}

TextureData LoadImageFile(const std::string& path)
{
    TextureData t = DecodeImage(read_file(path));
    t.Uri = path;
    return t;
}

} // namespace ptamd
