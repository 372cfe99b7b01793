"""Independent float64 restatements, written from the GLSL text and not from oracle/, of the three parts of the
hot path that nothing else pins: the direction generation of DisneySample (Disney.glsl:150-250 with
Sampling.glsl:34-47,70-88,147-156,179-184 and GLSL's reflect / refract), the payload of ClosestHit.glsl:144-226,
and the TracePath loop (RayGen.glsl:106-176) over a small scene with a brute-force Moller-Trumbore intersector.
The CPU oracle (fp32, its own intersector, its own code) must agree with them.

Tolerances: the restatements compute in float64 from the same fp32 inputs and the same RNG values, so they
differ from the oracle by fp32 rounding only: 2e-4 absolute on unit directions (the microfacet maps amplify
rounding by about 1/roughness), 1e-5 relative on payload fields, 1e-4 relative per sample on path radiance.
A sample is left out when a discrete decision (lobe choice, Fresnel choice, which triangle is hit) sits within
rounding distance of its threshold, because fp32 and float64 may then legitimately decide differently; the
tests assert that this happens rarely.
"""
import math

import numpy as np
import pytest

from pathtracer_amd.scene import Params, SceneData, make_material
from tests import oracle_lib
from tests.conftest import random_frames, random_surfaces
from tests.test_oracle_kat import np_disney_eval, py_pcg

PI = math.pi


# ---- Globals.h:61-64 ------------------------------------------------------------------------------------------------
def rand(seed):
    """RandomValue: float(hash) / 4294967295.0 in fp32 (the divisor rounds to 2^32), returned as a Python float."""
    h, seed = py_pcg(seed)
    return float(np.float32(h)) / 4294967296.0, seed


def norm(v):
    return v / math.sqrt(float(v @ v))


# ---- GLSL built-ins -------------------------------------------------------------------------------------------------
def reflect(I, N):
    return I - 2.0 * float(N @ I) * N


def refract(I, N, eta):
    k = 1.0 - eta * eta * (1.0 - float(N @ I) ** 2)
    if k < 0.0:
        return np.zeros(3)
    return eta * I - (eta * float(N @ I) + math.sqrt(k)) * N


def mix(a, b, t):
    return a * (1.0 - t) + b * t


def lum(c):
    return 0.212671 * c[0] + 0.715160 * c[1] + 0.072169 * c[2]


def schlick(u):
    m = min(max(1.0 - u, 0.0), 1.0)
    return m ** 5


def onb(N):                                                     # Sampling.glsl:179-184
    up = np.array([0.0, 0.0, 1.0]) if abs(N[2]) < 0.9999999 else np.array([1.0, 0.0, 0.0])
    T = norm(np.cross(up, N))
    return T, np.cross(N, T)


def dielectric_fresnel(c, eta):                                 # Sampling.glsl:131-145
    st = eta * eta * (1.0 - c * c)
    if st > 1.0:
        return 1.0
    ct = math.sqrt(max(1.0 - st, 0.0))
    rs = (eta * ct - c) / (eta * ct + c)
    rp = (eta * c - ct) / (eta * c + ct)
    return 0.5 * (rs * rs + rp * rp)


def sample_ggx_vndf(V, ax, ay, r1, r2):                         # Sampling.glsl:70-88
    Vh = norm(np.array([ax * V[0], ay * V[1], V[2]]))
    lensq = Vh[0] ** 2 + Vh[1] ** 2
    T1 = np.array([-Vh[1], Vh[0], 0.0]) / math.sqrt(lensq) if lensq > 0 else np.array([1.0, 0.0, 0.0])
    T2 = np.cross(Vh, T1)
    r = math.sqrt(r1)
    phi = 2.0 * PI * r2
    t1, t2 = r * math.cos(phi), r * math.sin(phi)
    s = 0.5 * (1.0 + Vh[2])
    t2 = (1.0 - s) * math.sqrt(1.0 - t1 * t1) + s * t2
    Nh = t1 * T1 + t2 * T2 + math.sqrt(max(0.0, 1.0 - t1 * t1 - t2 * t2)) * Vh
    return norm(np.array([ax * Nh[0], ay * Nh[1], max(0.0, Nh[2])]))


def sample_gtr1(rgh, r1, r2):                                   # Sampling.glsl:34-47
    a = max(0.001, rgh)
    a2 = a * a
    phi = r1 * 2.0 * PI
    cos_t = math.sqrt((1.0 - a2 ** (1.0 - r2)) / (1.0 - a2))
    sin_t = min(max(math.sqrt(max(1.0 - cos_t * cos_t, 0.0)), 0.0), 1.0)
    return np.array([sin_t * math.cos(phi), sin_t * math.sin(phi), cos_t])


def cosine_hemisphere(r1, r2):                                  # Sampling.glsl:147-156
    r = math.sqrt(r1)
    phi = 2.0 * PI * r2
    x, y = r * math.cos(phi), r * math.sin(phi)
    return np.array([x, y, math.sqrt(max(0.0, 1.0 - x * x - y * y))])


def np_disney_sample_direction(s, V, N, seed):
    """Disney.glsl:150-247 -> (L world, lobe name, margin of the closest discrete decision, advanced seed).
    s: the 17 surface floats of tests.conftest.random_surfaces."""
    alb = s[0:3].astype(float)
    metallic, spec_tint, cc, cc_rough, spec_trans = float(s[3]), float(s[6]), float(s[9]), float(s[10]), float(s[11])
    ax, ay, eta = float(s[13]), float(s[14]), float(s[15])
    V, N = V.astype(float), N.astype(float)
    r1, seed = rand(seed)
    r2, seed = rand(seed)
    T, B = onb(N)
    Vl = np.array([V @ T, V @ B, V @ N])
    l = lum(alb)
    ctint = alb / l if l > 0 else np.ones(3)
    F0 = ((1.0 - eta) / (1.0 + eta)) ** 2
    Cspec0 = F0 * mix(np.ones(3), ctint, spec_tint)
    diel_wt, metal_wt, glass_wt = (1 - metallic) * (1 - spec_trans), metallic, (1 - metallic) * spec_trans
    sw = schlick(Vl[2])
    pr = np.array([diel_wt * lum(alb), diel_wt * lum(mix(Cspec0, np.ones(3), sw)), metal_wt * lum(mix(alb, np.ones(3), sw)),
                   glass_wt, 0.25 * cc])
    pr = pr / pr.sum()
    cdf = np.cumsum(pr)
    r3, seed = rand(seed)
    margin = min(abs(r3 - cdf[0]), abs(r3 - cdf[2]), abs(r3 - cdf[3]))
    if r3 < cdf[0]:
        lobe, L = "diffuse", cosine_hemisphere(r1, r2)
    elif r3 < cdf[2]:
        H = sample_ggx_vndf(Vl, ax, ay, r1, r2)
        if H[2] < 0:
            H = -H
        lobe, L = "reflect", norm(reflect(-Vl, H))
    elif r3 < cdf[3]:
        H = sample_ggx_vndf(Vl, ax, ay, r1, r2)
        F = dielectric_fresnel(abs(float(Vl @ H)), eta)
        if H[2] < 0:
            H = -H
        r3b = (r3 - cdf[2]) / (cdf[3] - cdf[2])
        margin = min(margin, abs(r3b - F))
        if r3b < F:
            lobe, L = "glass-reflect", norm(reflect(-Vl, H))
        else:
            lobe, L = "glass-refract", norm(refract(-Vl, H, eta))
    else:
        H = sample_gtr1(cc_rough, r1, r2)
        if H[2] < 0:
            H = -H
        lobe, L = "clearcoat", norm(reflect(-Vl, H))
    return L[0] * T + L[1] * B + L[2] * N, lobe, margin, seed


@pytest.mark.parametrize("kind", ["diffuse", "metal", "mixed", "full", "glass"])
def test_disney_sample_direction_against_float64_restatement(kind):
    rng = np.random.default_rng(21)
    n = 600
    s = random_surfaces(rng, n, kind)
    s[:, 4] = np.maximum(s[:, 4], 0.05)
    aspect = np.sqrt(np.float32(1) - s[:, 16] * np.float32(0.9)).astype(np.float32)
    s[:, 13], s[:, 14] = np.maximum(np.float32(0.001), s[:, 4] / aspect), np.maximum(np.float32(0.001), s[:, 4] * aspect)
    V, N = random_frames(rng, n)
    seeds = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    f, L, pdf, seeds_out = oracle_lib.disney_sample(s, V, N, seeds)
    lobes, skipped = {}, 0
    for i in range(n):
        want, lobe, margin, seed = np_disney_sample_direction(s[i], V[i], N[i], int(seeds[i]))
        assert seed == int(seeds_out[i])
        if margin < 1e-5:
            skipped += 1
            continue
        lobes[lobe] = lobes.get(lobe, 0) + 1
        assert np.allclose(L[i], want, rtol=0, atol=2e-4), (i, lobe, L[i], want)
        # and the value returned with it is DisneyEval of that direction (Disney.glsl:249)
        wf, wp = np_disney_eval(s[i], V[i], N[i], want)
        if s[i, 4] >= 0.15 and wp > 1e-3:
            assert np.isclose(pdf[i], wp, rtol=2e-2, atol=1e-4), (i, lobe, pdf[i], wp)
    assert skipped <= n // 50
    need = {"diffuse": ["diffuse", "reflect"], "metal": ["reflect"], "mixed": ["diffuse", "reflect"],
            "full": ["diffuse", "reflect", "glass-reflect", "glass-refract", "clearcoat"], "glass": ["glass-reflect", "glass-refract"]}[kind]
    for lobe in need:
        assert lobes.get(lobe, 0) >= 5, (kind, lobes)


def test_total_internal_reflection_always_reflects():
    """Inside glass (eta = ior) at a grazing angle DielectricFresnel returns 1 (Sampling.glsl:135-137), so the rescaled
    r3 < F always holds and refract()'s zero vector (-> normalize(0) = NaN) is never produced."""
    s = np.zeros((64, 17), np.float32)
    s[:, 0:3] = 0.9
    s[:, 4] = 0.05
    s[:, 11] = 1.0
    s[:, 12] = s[:, 15] = 1.5
    s[:, 13] = s[:, 14] = 0.05
    N = np.tile(np.array([[0, 0, 1]], np.float32), (64, 1))
    V = np.tile(np.array([[0.9, 0.0, 0.3]], np.float32), (64, 1))
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    f, L, pdf, _ = oracle_lib.disney_sample(s, V, N, np.arange(1, 65))
    assert np.isfinite(L).all() and (L[:, 2] > 0).all()
    for i in range(64):
        want, lobe, _, _ = np_disney_sample_direction(s[i], V[i], N[i], i + 1)
        assert lobe == "glass-reflect" and np.allclose(L[i], want, atol=2e-4)


# ---- ClosestHit.glsl:144-226 ----------------------------------------------------------------------------------------
def np_closest_hit(sc, textures_rgb, submesh, prim, u, v, t, direction):
    """Payload fields in float64.  textures_rgb: per texture the (uniform) texel colour in [0, 1]."""
    _, vertex_offset, index_offset, material_index = [int(x) for x in sc.submeshes[submesh]]
    M = sc.materials[material_index]
    Mi = M.view(np.int32)
    idx = [int(sc.indices[prim * 3 + k + index_offset]) + vertex_offset for k in range(3)]
    bary = [1.0 - u - v, u, v]
    uv = np.zeros(2)
    nrm = np.zeros(3)
    for k in range(3):
        vert = sc.vertices[idx[k]].astype(float)
        uv += vert[3:5] * bary[k]
        nrm += vert[5:8] * bary[k]
    nrm = norm(nrm)
    o2w = sc.xforms[submesh].astype(float).reshape(3, 4)
    world_normal = norm(o2w[:, :3] @ nrm)                        # mat3(gl_ObjectToWorldEXT) * N, ClosestHit.glsl:170
    view = norm(-np.asarray(direction, float))
    albedo_tex = np.ones(3)
    if Mi[10] != -1:
        albedo_tex = np.asarray(textures_rgb[Mi[10]], float)
    mr = np.ones(2)
    if Mi[11] != -1:
        c = np.asarray(textures_rgb[Mi[11]], float)
        mr = np.array([c[2], c[1]])                              # .bg, ClosestHit.glsl:182
    p = {"Distance": t}
    alb = M[0:3].astype(float) * albedo_tex
    p["Albedo.x"], p["Albedo.y"], p["Albedo.z"] = alb
    p["Roughness"] = float(M[4]) * mr[1]
    p["Metallic"] = float(M[3]) * mr[0]
    em = M[5:8].astype(float) * float(M[8])
    p["Emission.x"], p["Emission.y"], p["Emission.z"] = em
    p["WorldNormal.x"], p["WorldNormal.y"], p["WorldNormal.z"] = world_normal
    for k, name in enumerate(["Anisotropic", "Subsurface", "SpecularTint", "Sheen", "SheenTint", "Clearcoat", "ClearcoatRoughness", "SpecTrans", "ior"]):
        p[name] = float(M[13 + k])
    aspect = math.sqrt(1.0 - p["Anisotropic"] * 0.9)
    p["ax"] = max(0.001, p["Roughness"] / aspect)
    p["ay"] = max(0.001, p["Roughness"] * aspect)
    p["eta"] = 1.0 / p["ior"] if float(view @ world_normal) < 0.0 else p["ior"]
    return p


def _one_triangle_scene():
    v = np.zeros((5, 12), np.float32)                            # two unused vertices in front: vertexOffset = 2
    v[2:, 0:3] = [(-1.0, -0.5, 0.2), (1.5, -0.25, -0.1), (0.25, 1.75, 0.4)]
    v[2:, 3:5] = [(0.1, 0.2), (0.9, 0.3), (0.4, 0.8)]
    v[2:, 5:8] = [(0.1, 0.2, 1.0), (-0.3, 0.1, 0.9), (0.2, -0.4, 0.8)]      # not unit length, like raw glTF data can be
    v[2:, 8:12] = (1, 0, 0, 1)
    idx = np.array([9, 9, 9, 0, 1, 2], np.uint32)                # indexOffset = 3
    xf = np.array([[2.0, 0.3, 0.0, 0.5, 0.1, 0.5, 0.2, -1.0, 0.0, -0.4, 1.5, 2.0]], np.float32)   # non-uniform scale + shear
    mats = [make_material(), make_material(albedo=(0.8, 0.5, 0.25), metallic=0.9, roughness=0.7, emissive=(0.5, 1.0, 2.0), emissive_strength=3.0,
                                           albedo_map=0, mr_map=1, anisotropic=0.6, subsurface=0.1, specular_tint=0.2, sheen=0.3, sheen_tint=0.4,
                                           clearcoat=0.5, clearcoat_roughness=0.15, spec_trans=0.25, ior=1.33)]
    tex_rgb = [(51 / 255, 153 / 255, 204 / 255), (25 / 255, 128 / 255, 230 / 255)]
    textures = [np.tile(np.array([round(c * 255) for c in rgb] + [255], np.uint8), (4, 4, 1)) for rgb in tex_rgb]
    sc = SceneData(v, idx, [(0, 2, 3, 1)], [3], xf, mats, textures=textures)
    return sc, tex_rgb


@pytest.mark.parametrize("direction", [(0.2, -0.1, -1.0), (-0.3, 0.2, 1.0), (0.0, 3.0, 0.1)])
def test_closest_hit_payload_against_float64_restatement(direction):
    sc, tex_rgb = _one_triangle_scene()
    o = oracle_lib.Oracle(sc, threads=1)
    etas = set()
    for (u, v) in [(0.2, 0.3), (0.0, 0.0), (0.65, 0.05), (0.1, 0.85)]:
        got = oracle_lib.closest_hit(o, 0, 0, u, v, 1.25, direction)
        want = np_closest_hit(sc, tex_rgb, 0, 0, np.float32(u).item(), np.float32(v).item(), 1.25, direction)
        for k, w in want.items():
            assert math.isclose(got[k], w, rel_tol=1e-5, abs_tol=1e-6), (k, got[k], w)
        etas.add(round(got["eta"], 4))
        # the swizzle and the anisotropy split did something
        assert math.isclose(got["Metallic"], 0.9 * 230 / 255, rel_tol=1e-5) and math.isclose(got["Roughness"], 0.7 * 128 / 255, rel_tol=1e-5)
        assert got["ax"] > got["ay"]
    assert len(etas) == 1
    o.close()


def test_closest_hit_eta_follows_the_side_the_ray_comes_from():
    sc, tex_rgb = _one_triangle_scene()
    o = oracle_lib.Oracle(sc, threads=1)
    n = np.array([oracle_lib.closest_hit(o, 0, 0, 0.3, 0.3, 1.0, (0, 0, -1))[k] for k in ("WorldNormal.x", "WorldNormal.y", "WorldNormal.z")])
    front = oracle_lib.closest_hit(o, 0, 0, 0.3, 0.3, 1.0, -n)["eta"]      # view = +n: dot >= 0 -> ior (ClosestHit.glsl:226)
    back = oracle_lib.closest_hit(o, 0, 0, 0.3, 0.3, 1.0, n)["eta"]
    assert math.isclose(front, 1.33, rel_tol=1e-6) and math.isclose(back, 1 / 1.33, rel_tol=1e-6)
    o.close()


# ---- RayGen.glsl:106-176 over a 16-triangle scene ----------------------------------------------------------------------
def _small_scene():
    """Open box (floor, back wall, two side walls, ceiling light) + a tilted glass pane + a clearcoated metal quad under a
    non-uniform transform: 16 triangles, 7 submeshes, 6 materials; constant white sky outside."""
    def quad(p0, p1, p2, p3, n):
        v = np.zeros((4, 12), np.float32)
        v[:, 0:3] = [p0, p1, p2, p3]
        v[:, 5:8] = n
        v[:, 8:12] = (1, 0, 0, 1)
        return v, np.array([0, 1, 2, 0, 2, 3], np.uint32)
    quads = [
        (quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), (0, 1, 0)), 0, None),                       # floor
        (quad((-1, 0, -1), (-1, 2, -1), (1, 2, -1), (1, 0, -1), (0, 0, 1)), 0, None),                     # back
        (quad((-1, 0, 1), (-1, 2, 1), (-1, 2, -1), (-1, 0, -1), (1, 0, 0)), 1, None),                     # left, red
        (quad((1, 0, -1), (1, 2, -1), (1, 2, 1), (1, 0, 1), (-1, 0, 0)), 2, None),                        # right, rough metal
        (quad((-0.5, 1.99, -0.5), (0.5, 1.99, -0.5), (0.5, 1.99, 0.5), (-0.5, 1.99, 0.5), (0, -1, 0)), 3, None),   # light
        (quad((-0.6, 0.3, 0.3), (0.2, 0.3, 0.5), (0.2, 1.2, 0.5), (-0.6, 1.2, 0.3), (-0.24, 0, 0.97)), 4, None),   # glass pane
        (quad((-0.5, 0, -0.5), (0.5, 0, -0.5), (0.5, 0, 0.5), (-0.5, 0, 0.5), (0, 1, 0)), 5,
         [0.6, 0.2, 0.0, 0.3, 0.0, 1.0, 0.3, 0.45, 0.1, -0.2, 0.5, -0.3]),                                 # sheared, raised quad
        (quad((-1, 2, -1), (-1, 2, 1), (1, 2, 1), (1, 2, -1), (0, -1, 0)), 0, None),                       # ceiling
    ]
    mats = [make_material(albedo=(0.75, 0.75, 0.75), roughness=1.0),
            make_material(albedo=(0.65, 0.06, 0.05), roughness=0.9),
            make_material(albedo=(0.9, 0.8, 0.5), metallic=1.0, roughness=0.45, anisotropic=0.4),
            make_material(albedo=(0.8, 0.8, 0.8), emissive=(1.0, 0.9, 0.7), emissive_strength=4.0),
            make_material(albedo=(0.9, 0.95, 1.0), roughness=0.35, spec_trans=1.0, ior=1.45),
            make_material(albedo=(0.2, 0.4, 0.8), metallic=0.3, roughness=0.5, clearcoat=1.0, clearcoat_roughness=0.3, sheen=0.5, sheen_tint=0.5, subsurface=0.3)]
    verts, inds, subs, counts, xfs = [], [], [], [], []
    nv = ni = 0
    ident = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]
    for (v, i), m, xf in quads:
        subs.append((0, nv, ni, m))
        counts.append(len(i))
        xfs.append(ident if xf is None else xf)
        verts.append(v)
        inds.append(i)
        nv += len(v)
        ni += len(i)
    return SceneData(np.concatenate(verts), np.concatenate(inds), subs, counts, np.array(xfs, np.float32), mats)


def np_world_triangles(sc):
    tris = []
    for s in range(len(sc.submeshes)):
        _, vo, io, _ = [int(x) for x in sc.submeshes[s]]
        m = sc.xforms[s].astype(float).reshape(3, 4)
        for p in range(int(sc.index_counts[s]) // 3):
            pts = [m[:, :3] @ sc.vertices[int(sc.indices[io + 3 * p + k]) + vo, 0:3].astype(float) + m[:, 3] for k in range(3)]
            tris.append((s, p, pts))
    return tris


def np_closest(tris, o, d, tmin, tmax):
    """Moller-Trumbore over every triangle: (t, u, v, submesh, prim) of the closest hit in (tmin, tmax), and how
    close the decision was (gap to the second-closest hit, distance of the hit to a triangle edge)."""
    best, second, edge = None, math.inf, math.inf
    for s, p, (a, b, c) in tris:
        e1, e2 = b - a, c - a
        pv = np.cross(d, e2)
        det = float(e1 @ pv)
        if det == 0.0:
            continue
        tv = o - a
        u = float(tv @ pv) / det
        qv = np.cross(tv, e1)
        v = float(d @ qv) / det
        t = float(e2 @ qv) / det
        if not (tmin < t < tmax):
            continue
        inside = min(u, v, 1.0 - u - v)
        if inside < 0.0:
            if inside > -1e-6:
                edge = min(edge, -inside)
            continue
        if best is None or t < best[0]:
            if best is not None:
                second = min(second, best[0])
            best = (t, u, v, s, p)
            best_edge = inside
        else:
            second = min(second, t)
    if best is None:
        return None, edge
    return best, min(edge, best_edge, second - best[0])


def np_trace_path(sc, tris, o, d, seed, max_bounces, eps=0.0003, tmin=0.00001, tmax=1e27, sky=10.0):
    """RayGen.glsl:106-176 in float64 -> (radiance, seed, smallest decision margin along the path)."""
    radiance, thr = np.zeros(3), np.ones(3)
    o, d = np.asarray(o, float), np.asarray(d, float)
    margin = math.inf
    for _ in range(max_bounces):
        hit, m = np_closest(tris, o, d, tmin, tmax)
        margin = min(margin, m)
        if hit is None:
            radiance += sky * thr                                # constant white environment x 10 (RayGen.glsl:132)
            break
        t, u, v, s, p = hit
        pl = np_closest_hit(sc, [], s, p, u, v, t, d)
        radiance += np.array([pl["Emission.x"], pl["Emission.y"], pl["Emission.z"]]) * thr
        N = np.array([pl["WorldNormal.x"], pl["WorldNormal.y"], pl["WorldNormal.z"]])
        ff = -N if float(-d @ N) < 0.0 else N
        surf = np.array([pl["Albedo.x"], pl["Albedo.y"], pl["Albedo.z"], pl["Metallic"], pl["Roughness"], pl["Subsurface"], pl["SpecularTint"],
                         pl["Sheen"], pl["SheenTint"], pl["Clearcoat"], pl["ClearcoatRoughness"], pl["SpecTrans"], pl["ior"], pl["ax"], pl["ay"],
                         pl["eta"], pl["Anisotropic"]])
        L, _, lm, seed = np_disney_sample_direction(surf, -d, ff, seed)
        margin = min(margin, lm)
        # DisneySample evaluates with V taken through the shading frame and back (Disney.glsl:246-249): same vector up to rounding
        f, pdf = np_disney_eval(surf, -d, ff, L)
        if pdf > 0.0:
            thr = thr * (f / pdf)
        else:
            break
        fhp = o + d * t
        d = L
        o = fhp + d * eps
    return radiance, seed, margin


def test_trace_path_against_float64_restatement():
    sc = _small_scene()
    tris = np_world_triangles(sc)
    assert len(tris) == 16
    B = 4
    o = oracle_lib.Oracle(sc, threads=1)
    o.resize(8, 8)
    o.set_params(Params.reference(spp_per_frame=1, max_bounces=B))
    rng = np.random.default_rng(33)
    eye = np.array([0.1, 1.0, 3.2])
    n, compared, lit, worst = 400, 0, 0, 0.0
    for i in range(n):
        target = np.array([rng.uniform(-0.95, 0.95), rng.uniform(0.02, 1.95), rng.uniform(-0.9, 0.9)])
        d = (target - eye).astype(np.float32)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        seed = int(rng.integers(0, 2 ** 32))
        got, seed_o = oracle_lib.trace_path(o, eye.astype(np.float32), d, seed, use_bvh=False)
        want, seed_n, margin = np_trace_path(sc, tris, eye.astype(np.float32), d, seed, B)
        if margin < 1e-4:
            continue                                             # a discrete decision too close to call in fp32
        compared += 1
        assert seed_o == seed_n, i
        err = np.abs(got - want) / (np.abs(want) + 1e-3)
        worst = max(worst, float(err.max()))
        assert (err < 1e-4).all(), (i, got, want, margin)
        lit += int(want.max() > 0)
        # the hierarchy changes nothing
        got_bvh, _ = oracle_lib.trace_path(o, eye.astype(np.float32), d, seed, use_bvh=True)
        assert np.array_equal(got_bvh, got)
    assert compared >= n * 0.85 and lit >= compared * 0.3, (compared, lit, worst)
    o.close()
