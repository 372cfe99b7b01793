// MeshSource: glTF 2.0 import into the tables the ray-tracing path reads.
//
// The reference loads its scenes with MeshSource(path) from its engine submodule
// (PathTracer/src/RayTracingLayer.cpp:27-32), whose source is not in the checkout; what the import
// must produce is fixed by the shaders instead: one shared vertex buffer of 48-byte vertices
// (ClosestHit.glsl:56-81), one uint32 index buffer, one submesh per drawn primitive with its vertex
// and index offsets and material index (ClosestHit.glsl:147-156), and 88-byte material records
// (ClosestHit.glsl:113-142).  Conventions where the engine is unknown: scene nodes are walked depth
// first in file order; glTF defaults for missing factors (base colour 1, metallic 1, roughness 1,
// emissive 0, emissive strength 1); KHR_materials_emissive_strength, _ior, _transmission, _clearcoat, _sheen and _anisotropy are read,
// ior defaults to 1.5 (the value Miss.glsl:28 uses), the other Disney parameters to 0; a missing
// tangent becomes (1,0,0,1); texture indices are glTF image indices.
#include "host.h"
#include "json.h"
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>

namespace ptamd {

namespace {

std::string read_file(const std::string& path, bool binary)
{
    std::ifstream f(path, binary ? std::ios::binary : std::ios::in);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

std::string base64_decode(const std::string& s, size_t start)
{
    static int8_t table[256];
    static bool init = false;
    if (!init) {
        std::memset(table, -1, sizeof(table));
        const char* a = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; i++) table[(unsigned char)a[i]] = (int8_t)i;
        init = true;
    }
    std::string out;
    out.reserve((s.size() - start) * 3 / 4);
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = start; i < s.size(); i++) {
        int8_t v = table[(unsigned char)s[i]];
        if (v < 0) continue;   // padding, whitespace
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out += (char)((acc >> bits) & 0xFF);
        }
    }
    return out;
}

struct Mat4d {
    double m[4][4];   // m[row][col]
    static Mat4d identity()
    {
        Mat4d r{};
        for (int i = 0; i < 4; i++) r.m[i][i] = 1.0;
        return r;
    }
    Mat4d operator*(const Mat4d& b) const
    {
        Mat4d r{};
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                double s = 0.0;
                for (int k = 0; k < 4; k++) s += m[i][k] * b.m[k][j];
                r.m[i][j] = s;
            }
        return r;
    }
};

Mat4d node_local(const Json& node)
{
    Mat4d r = Mat4d::identity();
    if (const Json* mj = node.find("matrix")) {
        for (int c = 0; c < 4; c++)
            for (int row = 0; row < 4; row++) r.m[row][c] = mj->at(c * 4 + row).num;
        return r;
    }
    double t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
    if (const Json* j = node.find("translation")) for (int i = 0; i < 3; i++) t[i] = j->at(i).num;
    if (const Json* j = node.find("rotation")) for (int i = 0; i < 4; i++) q[i] = j->at(i).num;
    if (const Json* j = node.find("scale")) for (int i = 0; i < 3; i++) s[i] = j->at(i).num;
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double rot[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                        {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                        {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) r.m[i][j] = rot[i][j] * s[j];
        r.m[i][3] = t[i];
    }
    return r;
}

struct Accessor {
    const uint8_t* base = nullptr;
    size_t count = 0, stride = 0;
    int componentType = 0, ncomp = 0;
    bool normalized = false;
};

size_t comp_size(int t)
{
    switch (t) {
    case 5120: case 5121: return 1;
    case 5122: case 5123: return 2;
    case 5125: case 5126: return 4;
    }
    throw std::runtime_error("gltf: unknown component type " + std::to_string(t));
}

int num_comp(const std::string& t)
{
    if (t == "SCALAR") return 1;
    if (t == "VEC2") return 2;
    if (t == "VEC3") return 3;
    if (t == "VEC4") return 4;
    if (t == "MAT4") return 16;
    throw std::runtime_error("gltf: unsupported accessor type " + t);
}

double read_comp(const Accessor& a, size_t i, int c)
{
    const uint8_t* p = a.base + i * a.stride + (size_t)c * comp_size(a.componentType);
    switch (a.componentType) {
    case 5126: { float f; std::memcpy(&f, p, 4); return f; }
    case 5125: { uint32_t v; std::memcpy(&v, p, 4); return v; }
    case 5123: { uint16_t v; std::memcpy(&v, p, 2); return a.normalized ? v / 65535.0 : v; }
    case 5122: { int16_t v; std::memcpy(&v, p, 2); return a.normalized ? std::fmax(v / 32767.0, -1.0) : v; }
    case 5121: { uint8_t v = *p; return a.normalized ? v / 255.0 : v; }
    case 5120: { int8_t v; std::memcpy(&v, p, 1); return a.normalized ? std::fmax(v / 127.0, -1.0) : v; }
    }
    return 0.0;
}

} // namespace

MeshSource::MeshSource(const std::string& path) : m_Path(path)
{
    std::string text = read_file(path, false);
    Json g = JsonParser(text).parse();
    std::string dir;
    size_t slash = path.find_last_of("/\\");
    if (slash != std::string::npos) dir = path.substr(0, slash + 1);

    // a non-negative integer field (offsets, counts, strides, indices into the file's arrays): anything negative,
    // fractional, non-finite or beyond 2^53 is refused before it can turn into a huge size_t
    auto field = [](const Json& o, const char* key, double dflt) -> size_t {
        double v = o.get(key, dflt);
        if (!(v >= 0.0) || v > 9007199254740992.0 || v != std::floor(v))
            throw std::runtime_error(std::string("gltf: '") + key + "' is not a non-negative integer");
        return (size_t)v;
    };
    auto index_of = [](double v, size_t n, const char* what) -> size_t {
        if (!(v >= 0.0) || v != std::floor(v) || v >= (double)n) throw std::runtime_error(std::string("gltf: ") + what + " index out of range");
        return (size_t)v;
    };

    std::vector<std::string> buffers;
    if (const Json* bj = g.find("buffers"))
        for (const Json& b : bj->arr) {
            std::string uri = b.gets("uri", "");
            if (uri.rfind("data:", 0) == 0) {
                size_t comma = uri.find(',');
                if (comma == std::string::npos) throw std::runtime_error("gltf: malformed data uri");
                buffers.push_back(base64_decode(uri, comma + 1));
            } else {
                buffers.push_back(read_file(dir + uri, true));
            }
            if (buffers.back().size() < field(b, "byteLength", 0))
                throw std::runtime_error("gltf: buffer '" + uri.substr(0, 40) + "' is shorter than its byteLength");
        }

    auto accessor = [&](double idx) {
        const Json& accs = g.at("accessors");
        const Json& a = accs.at(index_of(idx, accs.size(), "accessor"));
        if (a.find("sparse")) throw std::runtime_error("gltf: sparse accessors are not supported");
        const Json& bvs = g.at("bufferViews");
        const Json& bv = bvs.at(index_of(a.get("bufferView", -1.0), bvs.size(), "bufferView"));
        Accessor r;
        r.componentType = (int)a.geti("componentType", 0);
        r.ncomp = num_comp(a.gets("type", ""));
        r.count = field(a, "count", 0);
        const Json* nj = a.find("normalized");
        r.normalized = nj && nj->is(Json::Bool) && nj->b;
        const size_t elem = comp_size(r.componentType) * (size_t)r.ncomp;
        r.stride = field(bv, "byteStride", 0);
        if (r.stride == 0) r.stride = elem;
        if (r.stride < elem) throw std::runtime_error("gltf: byteStride is smaller than the element");
        const size_t off_bv = field(bv, "byteOffset", 0), off_a = field(a, "byteOffset", 0);
        const std::string& buf = buffers.at(index_of(bv.get("buffer", 0.0), buffers.size(), "buffer"));
        const size_t size = buf.size();
        // overflow-safe: every term is checked against what is left of the buffer
        if (off_bv > size || off_a > size - off_bv) throw std::runtime_error("gltf: accessor exceeds its buffer");
        const size_t off = off_bv + off_a;
        if (r.count) {
            if (elem > size - off || r.count - 1 > (size - off - elem) / r.stride) throw std::runtime_error("gltf: accessor exceeds its buffer");
        }
        r.base = reinterpret_cast<const uint8_t*>(buf.data()) + off;
        return r;
    };

    if (const Json* ij = g.find("images"))
        for (const Json& im : ij->arr) {
            TextureData t;
            t.Uri = im.gets("uri", "");
            m_Textures.push_back(std::move(t));
        }
    // sampler state travels with the image: wrapS / wrapT 10497 repeat (default), 33071 clamp to edge, 33648 mirrored
    // repeat; magFilter 9728 nearest, otherwise linear.  Mip-mapping (minFilter) has no counterpart: level 0 only.
    if (const Json* tj = g.find("textures")) {
        const Json* sj = g.find("samplers");
        std::vector<bool> seen(m_Textures.size(), false);
        auto wrap = [](int64_t v) -> uint32_t { return v == 33071 ? 1u : (v == 33648 ? 2u : 0u); };
        for (const Json& t : tj->arr) {
            int64_t src = t.geti("source", -1), si = t.geti("sampler", -1);
            if (src < 0 || (size_t)src >= m_Textures.size() || seen[(size_t)src]) continue;
            seen[(size_t)src] = true;
            if (!sj || si < 0 || (size_t)si >= sj->size()) continue;
            const Json& sm = sj->at((size_t)si);
            m_Textures[(size_t)src].SamplerFlags = wrap(sm.geti("wrapS", 10497)) | (wrap(sm.geti("wrapT", 10497)) << 2) |
                                                   (sm.geti("magFilter", 9729) == 9728 ? 16u : 0u);
        }
    }
    auto image_of = [&](const Json* ref) -> int32_t {
        if (!ref) return -1;
        int64_t ti = ref->geti("index", -1);
        const Json* tj = g.find("textures");
        if (ti < 0 || !tj || (size_t)ti >= tj->size()) return -1;
        return (int32_t)tj->at((size_t)ti).geti("source", -1);
    };

    if (const Json* mj = g.find("materials"))
        for (const Json& m : mj->arr) {
            MaterialData d{};
            d.AlbedoValue[0] = d.AlbedoValue[1] = d.AlbedoValue[2] = 1.0f;
            d.MetallicValue = 1.0f;
            d.RoughnessValue = 1.0f;
            d.EmissiveStrength = 1.0f;
            d.AlbedoMapIndex = d.MetallicRoughnessMapIndex = d.NormalMapIndex = -1;
            d.ior = 1.5f;
            if (const Json* pbr = m.find("pbrMetallicRoughness")) {
                if (const Json* c = pbr->find("baseColorFactor")) for (int i = 0; i < 3; i++) d.AlbedoValue[i] = (float)c->at(i).num;
                d.MetallicValue = (float)pbr->get("metallicFactor", 1.0);
                d.RoughnessValue = (float)pbr->get("roughnessFactor", 1.0);
                d.AlbedoMapIndex = image_of(pbr->find("baseColorTexture"));
                d.MetallicRoughnessMapIndex = image_of(pbr->find("metallicRoughnessTexture"));
            }
            if (const Json* e = m.find("emissiveFactor")) for (int i = 0; i < 3; i++) d.EmissiveValue[i] = (float)e->at(i).num;
            d.NormalMapIndex = image_of(m.find("normalTexture"));
            d.UseNormalMap = 0;
            if (const Json* ext = m.find("extensions")) {
                if (const Json* es = ext->find("KHR_materials_emissive_strength")) d.EmissiveStrength = (float)es->get("emissiveStrength", 1.0);
                if (const Json* io = ext->find("KHR_materials_ior")) d.ior = (float)io->get("ior", 1.5);
                // the glTF material extensions that have a counterpart among the shader's Disney parameters
                if (const Json* tr = ext->find("KHR_materials_transmission")) d.SpecTrans = (float)tr->get("transmissionFactor", 0.0);
                if (const Json* cc = ext->find("KHR_materials_clearcoat")) {
                    d.Clearcoat = (float)cc->get("clearcoatFactor", 0.0);
                    d.ClearcoatRoughness = (float)cc->get("clearcoatRoughnessFactor", 0.0);
                }
                if (const Json* sh = ext->find("KHR_materials_sheen")) {
                    float c[3] = {0.0f, 0.0f, 0.0f};
                    if (const Json* sc = sh->find("sheenColorFactor")) for (int i = 0; i < 3; i++) c[i] = (float)sc->at(i).num;
                    float hi = c[0] > c[1] ? (c[0] > c[2] ? c[0] : c[2]) : (c[1] > c[2] ? c[1] : c[2]);
                    float lo = c[0] < c[1] ? (c[0] < c[2] ? c[0] : c[2]) : (c[1] < c[2] ? c[1] : c[2]);
                    d.Sheen = hi;                                  // strength = brightest channel
                    d.SheenTint = hi > 0.0f ? (hi - lo) / hi : 0.0f;   // 0 for a grey sheen colour, 1 for a saturated one
                }
                if (const Json* an = ext->find("KHR_materials_anisotropy")) d.Anisotropic = (float)an->get("anisotropyStrength", 0.0);
            }
            m_Materials.push_back(d);
        }
    int32_t default_material = -1;

    struct Walker {
        MeshSource& self;
        const Json& g;
        decltype(accessor)& acc;
        int32_t& default_material;
        std::vector<char> on_path;      // nodes between the root and the current one: a node that names an ancestor is a cycle
        void walk(double nd, const Mat4d& parent, int depth)
        {
            const Json& nodes = g.at("nodes");
            if (!(nd >= 0.0) || nd != std::floor(nd) || nd >= (double)nodes.size()) throw std::runtime_error("gltf: node index out of range");
            const size_t ni = (size_t)nd;
            if (on_path.size() < nodes.size()) on_path.resize(nodes.size(), 0);
            if (on_path[ni]) throw std::runtime_error("gltf: the node hierarchy has a cycle");
            if (depth > 256) throw std::runtime_error("gltf: the node hierarchy is deeper than 256 levels");
            on_path[ni] = 1;
            const Json& node = nodes.at(ni);
            Mat4d world = parent * node_local(node);
            if (const Json* mi = node.find("mesh")) {
                const Json& meshes = g.at("meshes");
                if (!(mi->num >= 0.0) || mi->num != std::floor(mi->num) || mi->num >= (double)meshes.size()) throw std::runtime_error("gltf: mesh index out of range");
                const Json& mesh = meshes.at((size_t)mi->num);
                const Json& prims = mesh.at("primitives");
                for (size_t pi = 0; pi < prims.size(); pi++) {
                    const Json& prim = prims.at(pi);
                    if (prim.geti("mode", 4) != 4) continue;
                    const Json& at = prim.at("attributes");
                    Accessor pos = acc(at.at("POSITION").num);
                    if (pos.ncomp != 3) throw std::runtime_error("gltf: POSITION must be VEC3");
                    size_t n = pos.count;
                    Accessor uv{}, nrm{}, tan{};
                    if (const Json* j = at.find("TEXCOORD_0")) uv = acc(j->num);
                    if (const Json* j = at.find("NORMAL")) nrm = acc(j->num);
                    if (const Json* j = at.find("TANGENT")) tan = acc(j->num);
                    // the loops below read 2, 3 and 4 components: an accessor of a narrower type would be read past its extent
                    if (uv.base && uv.ncomp < 2) throw std::runtime_error("gltf: TEXCOORD_0 must have at least 2 components");
                    if (nrm.base && nrm.ncomp < 3) throw std::runtime_error("gltf: NORMAL must be VEC3");
                    if (tan.base && tan.ncomp < 4) throw std::runtime_error("gltf: TANGENT must be VEC4");
                    if (n > 0x7FFFFFFFu || self.m_Vertices.size() + n > 0x7FFFFFFFu) throw std::runtime_error("gltf: more than 2^31 vertices");
                    SubMesh sm;
                    sm.VertexOffset = (uint32_t)self.m_Vertices.size();
                    sm.IndexOffset = (uint32_t)self.m_Indices.size();
                    sm.VertexCount = (uint32_t)n;
                    for (size_t i = 0; i < n; i++) {
                        Vertex v{};
                        for (int c = 0; c < 3; c++) v.Position[c] = (float)read_comp(pos, i, c);
                        if (uv.base && i < uv.count) for (int c = 0; c < 2; c++) v.TextureCoords[c] = (float)read_comp(uv, i, c);
                        if (nrm.base && i < nrm.count) for (int c = 0; c < 3; c++) v.Normal[c] = (float)read_comp(nrm, i, c);
                        if (tan.base && i < tan.count) for (int c = 0; c < 4; c++) v.Tangent[c] = (float)read_comp(tan, i, c);
                        else { v.Tangent[0] = 1.0f; v.Tangent[3] = 1.0f; }
                        self.m_Vertices.push_back(v);
                    }
                    size_t ni_count;
                    if (const Json* ij = prim.find("indices")) {
                        Accessor ia = acc(ij->num);
                        if (ia.ncomp != 1) throw std::runtime_error("gltf: indices must be SCALAR");
                        if (ia.count > 0xFFFFFFFFu - self.m_Indices.size()) throw std::runtime_error("gltf: more than 2^32 indices");
                        ni_count = ia.count;
                        for (size_t i = 0; i < ia.count; i++) {
                            double v = read_comp(ia, i, 0);
                            if (v < 0 || v >= (double)n) throw std::runtime_error("gltf: index out of range");
                            self.m_Indices.push_back((uint32_t)v);
                        }
                    } else {
                        ni_count = n;
                        for (size_t i = 0; i < n; i++) self.m_Indices.push_back((uint32_t)i);
                    }
                    sm.IndexCount = (uint32_t)(ni_count - ni_count % 3);
                    int64_t mat = prim.geti("material", -1);
                    if (mat < 0 || (size_t)mat >= self.m_Materials.size()) {
                        if (default_material < 0) {
                            MaterialData d{};
                            d.AlbedoValue[0] = d.AlbedoValue[1] = d.AlbedoValue[2] = 1.0f;
                            d.MetallicValue = 0.0f;
                            d.RoughnessValue = 1.0f;
                            d.EmissiveStrength = 1.0f;
                            d.AlbedoMapIndex = d.MetallicRoughnessMapIndex = d.NormalMapIndex = -1;
                            d.ior = 1.5f;
                            default_material = (int32_t)self.m_Materials.size();
                            self.m_Materials.push_back(d);
                        }
                        mat = default_material;
                    }
                    sm.MaterialIndex = (uint32_t)mat;
                    std::string name = node.gets("name", mesh.gets("name", "node" + std::to_string(ni)));
                    sm.Name = pi == 0 ? name : name + "." + std::to_string(pi);
                    for (int c = 0; c < 4; c++)
                        for (int r = 0; r < 4; r++) sm.WorldTransform.m[c * 4 + r] = (float)world.m[r][c];
                    self.m_SubMeshes.push_back(sm);
                }
            }
            if (const Json* ch = node.find("children"))
                for (const Json& c : ch->arr) walk(c.num, world, depth + 1);
            on_path[ni] = 0;
        }
    } walker{*this, g, accessor, default_material, {}};

    const Json& scenes = g.at("scenes");
    const Json& roots = scenes.at(index_of(g.get("scene", 0.0), scenes.size(), "scene")).at("nodes");
    for (const Json& r : roots.arr) walker.walk(r.num, Mat4d::identity(), 0);
    if (m_SubMeshes.empty()) throw std::runtime_error("gltf: no triangle primitives in " + path);
}

size_t MeshSource::LoadTextures()
{
    std::string dir;
    size_t slash = m_Path.find_last_of("/\\");
    if (slash != std::string::npos) dir = m_Path.substr(0, slash + 1);
    size_t n = 0;
    for (TextureData& t : m_Textures) {
        if (!t.RGBA8.empty() || t.Uri.empty() || t.Uri.compare(0, 5, "data:") == 0) continue;
        std::string uri = t.Uri;
        for (size_t k = 0; k + 2 < uri.size(); k++)           // %20 and friends
            if (uri[k] == '%' && isxdigit((unsigned char)uri[k + 1]) && isxdigit((unsigned char)uri[k + 2])) {
                uri.replace(k, 3, 1, (char)std::stoi(uri.substr(k + 1, 2), nullptr, 16));
            }
        TextureData d = LoadImageFile(dir + uri);
        t.Width = d.Width; t.Height = d.Height; t.RGBA8 = std::move(d.RGBA8);
        n++;
    }
    return n;
}

} // namespace ptamd
