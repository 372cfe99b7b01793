"""Known-answer and property tests that pin the CPU oracle (no GPU).

The reference ships no tests or golden vectors, so every expected value here is derived by hand from
the shader text (cited per test) or computed by an independent float64 numpy restatement below.
"""
import json
import math
import os

import numpy as np
import pytest

from pathtracer_amd import camera
from pathtracer_amd.scene import Params, SceneData, make_material
from tests import oracle_lib
from tests.conftest import GOLDEN, random_frames, random_rays, random_surfaces

M32 = 0xFFFFFFFF


def py_pcg(seed):
    """Globals.h:54-59 in Python integers."""
    seed = (seed * 747796405 + 2891336453) & M32
    result = ((((seed >> ((seed >> 28) + 4)) ^ seed) & M32) * 277803737) & M32
    return ((result >> 22) ^ result) & M32, seed


def test_pcg_sequence_matches_hand_computation():
    for s0 in (0, 1, 12345, 0xFFFFFFFF):
        s_py = s_c = s0
        for _ in range(16):
            v_py, s_py = py_pcg(s_py)
            v_c, s_c = oracle_lib.pcg_hash(s_c)
            assert (v_py, s_py) == (v_c, s_c)


def test_pcg_golden_file():
    kat = json.load(open(os.path.join(GOLDEN, "pcg_kat.json")))
    for s0, seq in kat.items():
        s = int(s0)
        for want in seq:
            v, s = oracle_lib.pcg_hash(s)
            assert v == want


def test_pcg_first_value_of_seed_zero():
    # seed 0: state = 2891336453; shift = (state >> 28) + 4 = 14; hand-evaluated
    state = 2891336453
    word = (((state >> 14) ^ state) * 277803737) & M32
    want = (word >> 22) ^ word
    assert oracle_lib.pcg_hash(0) == (want, state)


def test_random_value_is_hash_times_2_pow_minus_32_and_can_be_one():
    # Globals.h:63 divides by the float literal 4294967295.0 = 2^32 in fp32
    for s0 in (0, 7, 99999):
        h, _ = py_pcg(s0)
        v, _ = oracle_lib.random_value(s0)
        assert v == np.float32(h) / np.float32(4294967296.0)
    # any hash >= 0xFFFFFF80 rounds to 2^32 as a float -> exactly 1.0
    assert np.float32(0xFFFFFF80) / np.float32(4294967296.0) == np.float32(1.0)


# ---- det_math accuracy ---------------------------------------------------------------------------
def ulp_err(got, want64):
    want32 = want64.astype(np.float32)
    ulp = np.spacing(np.abs(want32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / np.maximum(ulp, 1e-300)


def test_det_sin_cos_accuracy():
    x = np.linspace(0, 2 * math.pi, 200001).astype(np.float32)
    x64 = x.astype(np.float64)
    for fn, ref in ((0, np.sin), (1, np.cos)):
        got = oracle_lib.det_math(fn, x)
        want = ref(x64)
        # absolute error near zeros of the function, ulp error elsewhere
        err = np.abs(got - want)
        assert err.max() < 2.5e-7
        big = np.abs(want) > 0.1
        assert ulp_err(got[big], want[big]).max() <= 2.0


def test_det_log_exp_pow_accuracy():
    rng = np.random.default_rng(0)
    x = np.exp(rng.uniform(-80, 80, 100000)).astype(np.float32)
    assert ulp_err(oracle_lib.det_math(2, x), np.log(x.astype(np.float64))).max() <= 1.5
    e = rng.uniform(-87, 88, 100000).astype(np.float32)
    assert ulp_err(oracle_lib.det_math(3, e), np.exp(e.astype(np.float64))).max() <= 2.0
    a = rng.uniform(1e-6, 1, 100000).astype(np.float32)
    b = rng.uniform(0, 1, 100000).astype(np.float32)
    got = oracle_lib.det_math(4, a, b)
    want = np.power(a.astype(np.float64), b.astype(np.float64))
    assert (np.abs(got - want) / want).max() < 2e-6


def test_det_math_special_values():
    f = np.float32
    assert oracle_lib.det_math(2, [0.0])[0] == -np.inf
    assert np.isnan(oracle_lib.det_math(2, [-1.0])[0])
    assert oracle_lib.det_math(2, [1.0])[0] == 0.0
    assert oracle_lib.det_math(3, [0.0])[0] == 1.0
    assert oracle_lib.det_math(3, [-200.0])[0] == 0.0
    assert oracle_lib.det_math(3, [100.0])[0] == np.inf
    assert oracle_lib.det_math(4, [0.0], [0.5])[0] == 0.0          # pow(0, 0.5): black albedo in the refraction lobe
    assert oracle_lib.det_math(4, [1.0], [0.5])[0] == 1.0
    assert oracle_lib.det_math(0, [0.0])[0] == 0.0 and oracle_lib.det_math(1, [0.0])[0] == 1.0
    sub = f(1e-40)
    assert abs(oracle_lib.det_math(2, [sub])[0] - math.log(float(sub))) < 1e-4


# ---- Disney BSDF: independent float64 restatement of Disney.glsl:252-359 ---------------------------
def np_disney_eval(s, V, N, L):
    alb, metallic, rough, subsurface, spec_tint, sheen, sheen_tint, cc, cc_rough, spec_trans, ior, ax, ay, eta = (
        s[0:3].astype(float), *[float(v) for v in s[3:16]])
    V, N, L = V.astype(float), N.astype(float), L.astype(float)
    up = np.array([0, 0, 1.0]) if abs(N[2]) < 0.9999999 else np.array([1.0, 0, 0])
    T = np.cross(up, N); T /= np.linalg.norm(T)
    B = np.cross(N, T)
    loc = lambda v: np.array([v @ T, v @ B, v @ N])   # noqa: E731
    V, L = loc(V), loc(L)
    H = L + V if L[2] > 0 else L + V * eta
    H = H / np.linalg.norm(H)
    if H[2] < 0:
        H = -H
    lum = lambda c: 0.212671 * c[0] + 0.715160 * c[1] + 0.072169 * c[2]   # noqa: E731
    schlick = lambda u: min(max(1 - u, 0.0), 1.0) ** 5   # noqa: E731
    mixf = lambda a, b, t: a * (1 - t) + b * t   # noqa: E731

    def fresnel(c, e):
        st = e * e * (1 - c * c)
        if st > 1:
            return 1.0
        ct = math.sqrt(max(1 - st, 0))
        rs = (e * ct - c) / (e * ct + c)
        rp = (e * c - ct) / (e * c + ct)
        return 0.5 * (rs * rs + rp * rp)
    ctint = alb / lum(alb) if lum(alb) > 0 else np.ones(3)
    F0 = ((1 - eta) / (1 + eta)) ** 2
    Cspec0 = F0 * mixf(np.ones(3), ctint, spec_tint)
    Csheen = mixf(np.ones(3), ctint, sheen_tint)
    dielWt, metalWt, glassWt = (1 - metallic) * (1 - spec_trans), metallic, (1 - metallic) * spec_trans
    sw = schlick(V[2])
    pr = np.array([dielWt * lum(alb), dielWt * lum(mixf(Cspec0, np.ones(3), sw)), metalWt * lum(mixf(alb, np.ones(3), sw)),
                   glassWt, 0.25 * cc])
    pr = pr / pr.sum()
    refl = L[2] * V[2] > 0
    VdH = abs(V @ H)
    f = np.zeros(3)
    pdf = 0.0
    gtr2 = lambda: 1 / (math.pi * ax * ay * ((H[0] / ax) ** 2 + (H[1] / ay) ** 2 + H[2] ** 2) ** 2)   # noqa: E731
    smith = lambda w: 2 * abs(w[2]) / (abs(w[2]) + math.sqrt((w[0] * ax) ** 2 + (w[1] * ay) ** 2 + w[2] ** 2))   # noqa: E731

    def micro_refl(F):
        if L[2] <= 0:
            return np.zeros(3), 0.0
        D, G1 = gtr2(), smith(V)
        return F * D * G1 * smith(L) / (4 * L[2] * V[2]), G1 * D / (4 * V[2])
    if pr[0] > 0 and refl:
        if L[2] > 0:
            LdH = L @ H
            Rr = 2 * rough * LdH * LdH
            FL, FV = schlick(L[2]), schlick(V[2])
            Fretro = Rr * (FL + FV + FL * FV * (Rr - 1))
            Fd = (1 - 0.5 * FL) * (1 - 0.5 * FV)
            Fss90 = 0.5 * Rr
            Fss = mixf(1, Fss90, FL) * mixf(1, Fss90, FV)
            ss = 1.25 * (Fss * (1 / (L[2] + V[2]) - 0.5) + 0.5)
            fd = alb / math.pi * mixf(Fd + Fretro, ss, subsurface) + schlick(LdH) * sheen * Csheen
            f += fd * dielWt
            pdf += L[2] / math.pi * pr[0]
    if pr[1] > 0 and refl:
        F = (fresnel(VdH, 1 / ior) - F0) / (1 - F0)
        a, b = micro_refl(mixf(Cspec0, np.ones(3), F))
        f += a * dielWt
        pdf += b * pr[1]
    if pr[2] > 0 and refl:
        a, b = micro_refl(mixf(alb, np.ones(3), schlick(VdH)))
        f += a * metalWt
        pdf += b * pr[2]
    if pr[3] > 0:
        F = fresnel(VdH, eta)
        if refl:
            a, b = micro_refl(np.full(3, F))
            f += a * glassWt
            pdf += b * pr[3] * F
        elif L[2] < 0:
            LdH, VdH2 = L @ H, V @ H
            D, G1 = gtr2(), smith(V)
            G2 = G1 * smith(L)
            denom = (LdH + VdH2 * eta) ** 2
            jac = abs(LdH) / denom
            f += np.sqrt(alb) * (1 - F) * D * G2 * abs(VdH2) * jac * eta * eta / abs(L[2] * V[2]) * glassWt
            pdf += G1 * max(0.0, VdH2) * D * jac / V[2] * pr[3] * (1 - F)
    if pr[4] > 0 and refl and L[2] > 0:
        VdH2 = V @ H
        Fc = mixf(0.04, 1.0, schlick(VdH2))
        a = cc_rough
        if a >= 1:
            D = 1 / math.pi
        else:
            a2 = a * a
            D = (a2 - 1) / (math.pi * math.log(a2) * (1 + (a2 - 1) * H[2] * H[2]))
        G = (2 * L[2] / (L[2] + math.sqrt(0.0625 + L[2] ** 2 - 0.0625 * L[2] ** 2))) * (2 * V[2] / (V[2] + math.sqrt(0.0625 + V[2] ** 2 - 0.0625 * V[2] ** 2)))
        f += np.full(3, Fc) * D * G * 0.25 * cc
        pdf += D * H[2] / (4 * VdH2) * pr[4]
    return f * abs(L[2]), pdf


@pytest.mark.parametrize("kind", ["diffuse", "metal", "mixed", "full", "glass"])
def test_disney_eval_against_float64_restatement(kind):
    rng = np.random.default_rng(11)
    n = 400
    s = random_surfaces(rng, n, kind)
    s[:, 4] = np.maximum(s[:, 4], 0.15)          # keep the lobes well conditioned for a relative comparison
    s[:, 10] = np.maximum(s[:, 10], 0.05)
    aspect = np.sqrt(1 - s[:, 16] * 0.9)
    s[:, 13], s[:, 14] = np.maximum(0.001, s[:, 4] / aspect), np.maximum(0.001, s[:, 4] * aspect)
    V, N = random_frames(rng, n)
    L = rng.normal(size=(n, 3)).astype(np.float32)
    L /= np.linalg.norm(L, axis=1, keepdims=True)
    f, pdf = oracle_lib.disney_eval(s, V, N, L)
    for i in range(n):
        wf, wp = np_disney_eval(s[i], V[i], N[i], L[i])
        assert np.allclose(f[i], wf, rtol=2e-3, atol=1e-5), (i, f[i], wf)
        assert np.isclose(pdf[i], wp, rtol=2e-3, atol=1e-5), (i, pdf[i], wp)


def test_disney_sample_returns_eval_of_its_direction_and_draws_three_numbers():
    rng = np.random.default_rng(5)
    n = 2000
    s = random_surfaces(rng, n, "full")
    s[:, 4] = np.maximum(s[:, 4], 0.3)           # smooth lobes amplify the rounding of the re-projection
    s[:, 10] = np.maximum(s[:, 10], 0.3)
    aspect = np.sqrt(1 - s[:, 16] * 0.9)
    s[:, 13], s[:, 14] = np.maximum(0.001, s[:, 4] / aspect), np.maximum(0.001, s[:, 4] * aspect)
    V, N = random_frames(rng, n)
    seeds = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    f, L, pdf, seeds_out = oracle_lib.disney_sample(s, V, N, seeds)
    # Disney.glsl:154-155,200: exactly three RandomValue() calls per sample
    for i in range(0, n, 97):
        st = int(seeds[i])
        for _ in range(3):
            _, st = py_pcg(st)
        assert st == int(seeds_out[i])
    ok = np.isfinite(L).all(1)
    f2, pdf2 = oracle_lib.disney_eval(s[ok], V[ok], N[ok], L[ok])
    # the shader re-projects V through the shading frame before evaluating (Disney.glsl:246-249),
    # so eval(V) differs from the sampled result by rounding only
    fin = np.isfinite(f[ok]).all(1) & np.isfinite(f2).all(1)
    rel = np.abs(f2[fin] - f[ok][fin]) / (np.abs(f[ok][fin]) + 1e-4)
    assert np.quantile(rel, 0.99) < 1e-2 and np.median(rel) < 1e-5
    relp = np.abs(pdf2[fin] - pdf[ok][fin]) / (np.abs(pdf[ok][fin]) + 1e-4)
    assert np.quantile(relp, 0.99) < 1e-2 and np.median(relp) < 1e-5


def test_disney_pdf_integrates_to_at_most_one():
    rng = np.random.default_rng(3)
    for kind in ("diffuse", "metal", "mixed"):
        s = random_surfaces(rng, 1, kind)
        s[0, 4] = 0.5
        s[0, 13] = s[0, 14] = 0.5
        V, N = random_frames(rng, 1)
        n = 40000
        L = rng.normal(size=(n, 3)).astype(np.float32)
        L /= np.linalg.norm(L, axis=1, keepdims=True)
        _, pdf = oracle_lib.disney_eval(np.repeat(s, n, 0), np.repeat(V, n, 0), np.repeat(N, n, 0), L)
        integral = pdf.astype(np.float64).mean() * 4 * math.pi
        # reflections of visible normals that point below the horizon carry no density: <= 1, not == 1
        assert 0.5 < integral < 1.05, (kind, integral)


def test_white_furnace_weight_is_bounded():
    rng = np.random.default_rng(4)
    n = 20000
    s = np.repeat(random_surfaces(rng, 1, "diffuse"), n, 0)
    s[:, 0:3] = 1.0
    V, N = random_frames(rng, 1)
    seeds = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    f, L, pdf, _ = oracle_lib.disney_sample(s, np.repeat(V, n, 0), np.repeat(N, n, 0), seeds)
    w = np.where(pdf[:, None] > 0, f / np.maximum(pdf[:, None], 1e-30), 0)
    assert w.mean() < 1.1          # a white surface reflects at most (about) all of the light


def test_shader_helper_known_answers():
    # SchlickWeight (Sampling.glsl:124-129), DielectricFresnel TIR (:135-137), GTR1(a>=1) = 1/pi (:27-28)
    # are exercised through DisneyEval on materials that isolate them.
    s = np.zeros((1, 17), np.float32)
    s[0, 0:3] = 1.0
    s[0, 4] = 1.0
    s[0, 12] = 1.0
    s[0, 13] = s[0, 14] = 1.0
    s[0, 15] = 1.0       # eta = 1: F0 = 0, Cspec0 = 0
    N = np.array([[0, 0, 1]], np.float32)
    V = np.array([[0, 0, 1]], np.float32)
    L = np.array([[0, 0, 1]], np.float32)
    f, pdf = oracle_lib.disney_eval(s, V, N, L)
    # at normal incidence every Schlick weight is 0: diffuse = albedo/pi * (Fd + Fretro) with Fd = 1, Rr = 2 -> Fretro = 0
    # dielectric lobe: F = (Fresnel(1, 1) - 0)/(1 - 0) = 0 -> no specular; only the diffuse probability survives
    assert np.allclose(f[0], 1 / math.pi, rtol=1e-6)
    assert np.isclose(pdf[0], 1 / math.pi, rtol=1e-6)


def test_onb_switches_axis_near_the_pole():
    # Sampling.glsl:181: up = z unless |N.z| >= 0.9999999 -> a normal exactly along z must still give finite results
    s = random_surfaces(np.random.default_rng(0), 1, "diffuse")
    N = np.array([[0, 0, 1]], np.float32)
    V = np.array([[0.3, 0.1, 0.9]], np.float32)
    V /= np.linalg.norm(V)
    f, L, pdf, _ = oracle_lib.disney_sample(s, V, N, [123])
    assert np.isfinite(f).all() and np.isfinite(L).all() and pdf[0] > 0


def test_all_lobe_weights_zero_ends_the_path():
    # black metal seen at normal incidence: every lobe weight is 0, invTotalWt = 1/0 (Disney.glsl:184) and the
    # probabilities become NaN; every "pr > 0" test in DisneyEval is then false, so f = 0 and pdf = 0 and
    # TracePath breaks (RayGen.glsl:149-152) -- no NaN reaches the pixel on this route
    s = np.zeros((1, 17), np.float32)
    s[0, 3] = 1.0
    s[0, 4] = 0.5
    s[0, 12] = 1.0
    s[0, 13] = s[0, 14] = 0.5
    s[0, 15] = 1.0
    N = V = np.array([[0, 0, 1]], np.float32)
    f, L, pdf, _ = oracle_lib.disney_sample(s, V, N, [1])
    assert pdf[0] == 0.0 and not f.any()


# ---- traversal ------------------------------------------------------------------------------------
def test_bvh_equals_brute_force_on_cornell(cornell):
    rng = np.random.default_rng(8)
    o = oracle_lib.Oracle(cornell)
    rays = random_rays(rng, 3000, [-0.3, 0.0, -0.58], [0.25, 0.53, 0.6])
    assert np.array_equal(o.intersect(rays, use_bvh=True), o.intersect(rays, use_bvh=False))


def test_suzanne_fixture_bvh_equals_brute_force_and_keeps_its_textures(tmp_path):
    """The textured reference asset (Suzanne.gltf + its two PNGs as committed data): hierarchy = brute force for random
    rays and for a whole small frame; the scene file round-trips with textures and sampler flags."""
    sc = SceneData.load(os.path.join(GOLDEN, "suzanne_scene.npz"))
    assert sc.num_triangles == 3936 and [t.shape for t in sc.textures] == [(1024, 1024, 4)] * 2 and sc.texture_flags == [0, 0]
    assert sc.materials.view(np.int32)[0, 10:13].tolist() == [0, 1, -1]
    o = oracle_lib.Oracle(sc)
    rays = random_rays(np.random.default_rng(9), 4000, [-1.5, -1.1, -1.0], [1.5, 1.1, 1.0])
    hb = o.intersect(rays, use_bvh=True)
    assert np.array_equal(hb, o.intersect(rays, use_bvh=False)) and (hb[:, 3] != M32).sum() > 500
    cam = np.load(os.path.join(GOLDEN, "suzanne_cameras.npz"))["96x54"]
    o.resize(96, 54)
    o.set_camera(cam)
    o.set_params(Params.reference(spp_per_frame=1, max_bounces=6))
    for f in (1, 2):
        o.trace(f, use_bvh=True)
    a, ids = o.read_accum(), o.read_primary_ids()
    for f in (1, 2):
        o.trace(f, use_bvh=False)
    assert np.array_equal(a.view(np.uint32), o.read_accum().view(np.uint32)) and np.array_equal(ids, o.read_primary_ids())
    assert (ids[..., 0] != M32).mean() > 0.2
    path = str(tmp_path / "s.npz")
    sc.texture_flags = [5, 16]
    sc.save(path)
    back = SceneData.load(path)
    assert back.texture_flags == [5, 16] and all(np.array_equal(x, y) for x, y in zip(back.textures, sc.textures))


def test_hit_respects_tmin_tmax_and_tie_rule():
    # two coincident quads (4 triangles in 2 submeshes) at z = -1: equal t -> lowest (submesh, prim) wins
    v = np.zeros((8, 12), np.float32)
    quad = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)]
    v[0:4, 0:3] = quad
    v[4:8, 0:3] = quad
    v[:, 5:8] = (0, 0, 1)
    idx = np.array([0, 1, 2, 0, 2, 3, 0, 1, 2, 0, 2, 3], np.uint32)
    sc = SceneData(v, idx, [(0, 4, 6, 0), (0, 0, 0, 0)], [6, 6], np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (2, 1)),
                   [make_material()])
    o = oracle_lib.Oracle(sc)
    ray = np.array([[0.3, -0.2, 0, 1e-5, 0, 0, -1, 1e27]], np.float32)
    for use_bvh in (False, True):
        h = o.intersect(ray, use_bvh)
        assert h[0, 0:1].view(np.float32)[0] == 1.0 and tuple(h[0, 3:5]) == (0, 0)
    # tmax at the surface excludes it (t < tmax), tmin at the surface excludes it (t > tmin)
    miss = np.array([[0.3, -0.2, 0, 1e-5, 0, 0, -1, 1.0]], np.float32)
    assert tuple(o.intersect(miss, True)[0, 3:5]) == (M32, M32)
    miss2 = np.array([[0.3, -0.2, 0, 1.0, 0, 0, -1, 1e27]], np.float32)
    assert tuple(o.intersect(miss2, True)[0, 3:5]) == (M32, M32)
    # a ray through the shared diagonal edge hits (watertight), and through a corner vertex too
    edge = np.array([[0.5, 0.5, 0, 1e-5, 0, 0, -1, 1e27], [1.0, 1.0, 0, 1e-5, 0, 0, -1, 1e27]], np.float32)
    h = o.intersect(edge, True)
    assert (h[:, 3] != M32).all()


# ---- frame semantics (RayGen.glsl:312-373) ----------------------------------------------------------
def small_oracle(cornell, W=24, H=16, **kw):
    o = oracle_lib.Oracle(cornell)
    o.resize(W, H)
    o.set_camera(camera.cornell_camera(W, H))
    o.set_params(Params.reference(**kw))
    return o


def test_frame_one_is_displayed_but_not_accumulated(cornell):
    o = small_oracle(cornell, spp_per_frame=2, max_bounces=3)
    o.trace(1)
    assert not o.read_accum().any()                 # RayGen.glsl:361-365
    img1 = o.read_image()
    assert img1[..., :3].max() > 0 and (img1[..., 3] == 1).all()
    o.trace(2)
    acc = o.read_accum()
    assert (acc[..., 3] == 2).all()                 # numPaths = 0 + SAMPLE_COUNT
    o.trace(3)
    acc3 = o.read_accum()
    assert (acc3[..., 3] == 4).all()
    img = o.read_image()
    ok = ~((img[..., 0] == 1) & (img[..., 1] == 0))
    assert np.allclose(img[..., :3][ok], (acc3[..., :3] / acc3[..., 3:4])[ok], rtol=1e-6)


def test_pixel_zero_has_seed_zero_every_frame(cornell):
    # seed = (x + y*W) * FrameIndex (RayGen.glsl:314-315): pixel (0,0) repeats the same sample every frame
    o = small_oracle(cornell, spp_per_frame=1, max_bounces=4)
    o.trace(2)
    a2 = o.read_accum()[0, 0].copy()
    o.trace(3)
    a3 = o.read_accum()[0, 0]
    assert np.allclose(a3[:3], 2 * a2[:3], rtol=1e-6) and a3[3] == 2


def test_ray_count_bounds(cornell):
    o = small_oracle(cornell, W=16, H=16, spp_per_frame=3, max_bounces=5)
    o.trace(1)
    n = o.ray_count()
    assert 16 * 16 * 3 <= n <= 16 * 16 * 3 * 5


def test_nan_radiance_is_painted_red(cornell):
    # emission = EmissiveValue * EmissiveStrength (ClosestHit.glsl:202) = 0 * inf = NaN on the floor material;
    # RayGen.glsl:369-370 then writes (1,0,0,1) for every pixel whose path touches it
    sc = SceneData(cornell.vertices, cornell.indices, cornell.submeshes, cornell.index_counts, cornell.xforms,
                   cornell.materials.copy())
    sc.materials[0] = make_material(albedo=(1, 1, 1), metallic=0.0, roughness=1.0, emissive=(0, 0, 0), emissive_strength=np.inf)
    o = oracle_lib.Oracle(sc)
    o.resize(16, 16)
    o.set_camera(camera.cornell_camera(16, 16))
    o.set_params(Params.reference(spp_per_frame=1, max_bounces=2))
    o.trace(1)
    img = o.read_image()
    red = (img == np.array([1, 0, 0, 1], np.float32)).all(-1)
    assert red.any() and not np.isnan(img).any()


def test_golden_cornell_primary_ids(cornell, cameras):
    want = np.load(os.path.join(GOLDEN, "cornell_primary_ids_64.npy"))
    o = oracle_lib.Oracle(cornell)
    o.resize(64, 64)
    o.set_camera(cameras["64x64"])
    o.set_params(Params.reference(spp_per_frame=1, max_bounces=1))
    o.trace(1, use_bvh=True)
    assert np.array_equal(o.read_primary_ids(), want)


def test_golden_cornell_radiance(cornell, cameras):
    want = np.load(os.path.join(GOLDEN, "cornell_radiance_128_16spp.npy"))
    o = oracle_lib.Oracle(cornell)
    o.resize(128, 128)
    o.set_camera(cameras["128x128"])
    o.set_params(Params.reference(spp_per_frame=1, max_bounces=4))
    for f in range(1, 18):
        o.trace(f)
    assert np.array_equal(o.read_accum().view(np.uint32), want.view(np.uint32))


# ---- compute passes around the path tracer (oracle/passes.h) --------------------------------------------------
def ulp_err(got, want64):
    got = np.asarray(got, np.float64)
    spacing = np.spacing(np.abs(want64).astype(np.float32)).astype(np.float64)
    return np.abs(got - want64) / spacing


def test_det_atan2_acos_tan_accuracy():
    rng = np.random.default_rng(5)
    y = rng.normal(size=200000).astype(np.float32) * np.exp(rng.uniform(-20, 20, 200000)).astype(np.float32)
    x = rng.normal(size=200000).astype(np.float32) * np.exp(rng.uniform(-20, 20, 200000)).astype(np.float32)
    got = oracle_lib.det_math(5, y, x)
    assert ulp_err(got, np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() <= 3.0
    # axes and signs
    for yy, xx, want in ((0.0, 1.0, 0.0), (1.0, 0.0, math.pi / 2), (0.0, -1.0, math.pi), (-1.0, 0.0, -math.pi / 2), (-0.0, -1.0, -math.pi),
                         (1.0, 1.0, math.pi / 4), (0.0, 0.0, 0.0)):
        assert abs(float(oracle_lib.det_math(5, [yy], [xx])[0]) - want) < 3e-7
    c = rng.uniform(-1, 1, 200000).astype(np.float32)
    got = oracle_lib.det_math(6, c)
    err = np.abs(got.astype(np.float64) - np.arccos(c.astype(np.float64)))
    assert err.max() < 6e-7                       # absolute: acos is ill-conditioned at +-1
    assert oracle_lib.det_math(6, [1.0, -1.0, 1.5, -7.0]).tolist() == pytest.approx([0.0, math.pi, 0.0, math.pi], abs=3e-7)
    t = rng.uniform(-1.5, 1.5, 100000).astype(np.float32)
    assert ulp_err(oracle_lib.det_math(7, t), np.tan(t.astype(np.float64))).max() <= 4.0


def test_round_to_half_equals_numpy_float16():
    rng = np.random.default_rng(6)
    x = (rng.normal(size=400000) * np.exp(rng.uniform(-30, 14, 400000))).astype(np.float32)
    x[:12] = [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001, np.inf, 6.1e-5]
    ties = (np.arange(2048, dtype=np.float32) + 1024.0) * np.float32(2.0 ** -10) + np.float32(2.0 ** -11)   # exact halfway cases
    x = np.concatenate([x, ties, -ties])
    got = oracle_lib.det_math(8, x)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def np_postprocess(rgb, exposure):
    """float64 restatement of PostProcessing.glsl:14-55, written from the shader text."""
    m1 = np.array([[0.59719, 0.07600, 0.02840], [0.35458, 0.90834, 0.13383], [0.04823, 0.01566, 0.83777]]).T   # columns as given
    m2 = np.array([[1.60475, -0.10208, -0.00327], [-0.53108, 1.10813, -0.07276], [-0.07367, -0.00605, 1.07602]]).T
    v = (rgb.astype(np.float64) * exposure) @ m1.T
    a = v * (v + 0.0245786) - 0.000090537
    b = v * (0.983729 * v + 0.4329510) + 0.238081
    c = np.clip((a / b) @ m2.T, 0.0, 1.0)
    return c ** (1.0 / 2.2)


def test_postprocess_against_float64_restatement():
    rng = np.random.default_rng(7)
    img = np.exp(rng.uniform(-8, 4, (64, 64, 4))).astype(np.float32)
    img[0, 0] = 0.0
    img[0, 1] = [1e6, 1e6, 1e6, 1.0]
    got = oracle_lib.postprocess(img, 0.8)
    want = np_postprocess(img[..., :3], np.float32(0.8)) * 255.0
    assert np.all(got[..., 3] == 255)
    assert np.abs(got[..., :3].astype(np.float64) - want).max() <= 0.5 + 2e-3      # rounding + float32 evaluation
    assert got[0, 0, :3].tolist() == [0, 0, 0]             # a/b < 0 at black: clamped
    assert got[0, 1, :3].min() >= 250
    # NaN and negative radiance do not crash and give defined bytes
    bad = np.array([[[np.nan, -1.0, np.inf, 1.0]]], np.float32)
    assert oracle_lib.postprocess(bad, 0.8).shape == (1, 1, 4)


def cube_dirs(n):
    """Direction of every cube texel per GetCubeMapTexCoord (EquirectangularToCubeMap.glsl:11-25), float64."""
    g = np.arange(n, dtype=np.float64) / n
    ux, uy = np.meshgrid(2.0 * g - 1.0, 2.0 * (1.0 - g) - 1.0)
    one = np.ones_like(ux)
    faces = [(one, uy, -ux), (-one, uy, ux), (ux, one, -uy), (ux, -one, uy), (ux, uy, one), (-ux, uy, -one)]
    d = np.stack([np.stack(f, -1) for f in faces])
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def test_equirect_to_cube_properties():
    n = 16
    const = np.tile(np.array([0.3, 1.7, 100.1, 1.0], np.float32), (8, 16, 1))
    cube = oracle_lib.equirect_to_cube(const, n)
    want = const[0, 0].astype(np.float16).astype(np.float32)          # the cube is rgba16f
    assert np.array_equal(cube.reshape(-1, 4), np.tile(want, (6 * n * n, 1)))
    # a smooth image of the direction: cube texel = image evaluated at its (u, v), up to filtering and binary16
    w, h = 512, 256
    u = (np.arange(w) + 0.5) / w
    v = (np.arange(h) + 0.5) / h
    img = np.zeros((h, w, 4), np.float32)
    img[..., 0] = np.cos(2 * np.pi * u)[None, :] * 0.5 + 1.0
    img[..., 1] = v[:, None]
    img[..., 2] = 2.0
    img[..., 3] = 1.0
    cube = oracle_lib.equirect_to_cube(img, n).astype(np.float64)
    d = cube_dirs(n)
    uu = np.arctan2(d[..., 2], d[..., 0]) / (2 * 3.141592) + 0.5
    vv = np.arccos(d[..., 1]) / 3.141592
    assert np.abs(cube[..., 0] - (np.cos(2 * np.pi * uu) * 0.5 + 1.0)).max() < 2e-3
    assert np.abs(cube[..., 1] - vv).max() < 2e-3
    assert np.all(cube[..., 2] == 2.0)
    # +Y face looks at the top row of the image, -Y at the bottom row
    assert cube[2, n // 2, n // 2, 1] < 0.1 and cube[3, n // 2, n // 2, 1] > 0.9


def np_preetham(d, turbidity, azimuth, inclination):
    """float64 restatement of PreethamSky.glsl:35-146 for unit directions d, written from the shader text."""
    PI = 3.14159265359
    t = turbidity
    s = np.array([np.sin(inclination) * np.cos(azimuth), np.cos(inclination), np.sin(inclination) * np.sin(azimuth)])
    s = s / np.linalg.norm(s)
    A = np.array([0.1787 * t - 1.4630, -0.0193 * t - 0.2592, -0.0167 * t - 0.2608])
    B = np.array([-0.3554 * t + 0.4275, -0.0665 * t + 0.0008, -0.0950 * t + 0.0092])
    Cc = np.array([-0.0227 * t + 5.3251, -0.0004 * t + 0.2125, -0.0079 * t + 0.2102])
    D = np.array([0.1206 * t - 2.5771, -0.0641 * t - 0.8989, -0.0441 * t - 1.6537])
    E = np.array([-0.0670 * t + 0.3703, -0.0033 * t + 0.0452, -0.0109 * t + 0.0529])
    thS = np.arccos(max(s[1], 0.0))
    thE = np.arccos(np.maximum(d[..., 1], 0.0))
    gam = np.arccos(np.clip(np.maximum(d @ s, 0.0), -1, 1))
    chi = (4.0 / 9.0 - t / 120.0) * (PI - 2.0 * thS)
    Yz = (4.0453 * t - 4.9710) * np.tan(chi) - 0.2155 * t + 2.4192
    th2, th3 = thS * thS, thS ** 3
    xz = (0.00165 * th3 - 0.00375 * th2 + 0.00209 * thS) * t * t + (-0.02903 * th3 + 0.06377 * th2 - 0.03202 * thS + 0.00394) * t + \
         (0.11693 * th3 - 0.21196 * th2 + 0.06052 * thS + 0.25886)
    yz = (0.00275 * th3 - 0.00610 * th2 + 0.00317 * thS) * t * t + (-0.04214 * th3 + 0.08970 * th2 - 0.04153 * thS + 0.00516) * t + \
         (0.15346 * th3 - 0.26756 * th2 + 0.06670 * thS + 0.26688)

    def perez(theta, gamma):
        theta, gamma = np.asarray(theta)[..., None], np.asarray(gamma)[..., None]
        with np.errstate(over="ignore", divide="ignore"):
            return (1.0 + A * np.exp(B / np.cos(theta))) * (1.0 + Cc * np.exp(D * gamma) + E * np.cos(gamma) ** 2)

    Yp = np.array([Yz, xz, yz]) * (perez(thE, gam) / perez(0.0, thS))
    Y, x, y = Yp[..., 0], Yp[..., 1], Yp[..., 2]
    X = x * (Y / y)
    Z = (1.0 - x - y) * (Y / y)
    M = np.array([[2.3706743, -0.9000405, -0.4706338], [-0.5138850, 1.4253036, 0.0885814], [0.0052982, -0.0146949, 1.0093968]])
    return np.stack([X, Y, Z], -1) @ M.T * 0.05


@pytest.mark.parametrize("params", [(3.14, 0.0, 0.0), (2.2, 1.0, 0.9), (6.0, 4.0, 1.3)])
def test_preetham_sky_against_float64_restatement(params):
    n = 24
    cube = oracle_lib.preetham_sky(*params, n)
    assert np.all(cube[..., 3] == 1.0)
    want = np_preetham(cube_dirs(n), *[float(np.float32(p)) for p in params])
    upper = cube_dirs(n)[..., 1] > 0.05            # below the horizon cos(theta) = 0: the shader divides by it
    got = cube[..., :3].astype(np.float64)
    scale = np.abs(want[upper]).max()
    assert np.isfinite(got[upper]).all()
    assert np.abs(got[upper] - want[upper]).max() < 2e-4 * scale


@pytest.mark.parametrize("ws,wt,nearest", [(0, 0, 0), (1, 1, 0), (2, 2, 0), (1, 2, 0), (2, 0, 1), (0, 1, 1)])
def test_texture_sampler_modes_against_the_vulkan_rules(ws, wt, nearest):
    """Wrap modes and filters of texture(sampler2D, uv) against a float64 restatement of the Vulkan texel rules:
    unnormalised coordinate u*w, bilinear taps at floor(x - 0.5) and the next texel (nearest: floor(x)), index i
    wrapped as i mod w (repeat), clamp(i, 0, w-1) (clamp to edge), mirror((i mod 2w)) (mirrored repeat)."""
    rng = np.random.default_rng(100 + ws + 4 * wt + 16 * nearest)
    tex = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
    uv = rng.uniform(-3, 3, (4000, 2)).astype(np.float32)
    uv[:8] = [[0, 0], [1, 1], [-1, 2], [0.5, 0.5], [0.999999, 1e-7], [2, -2], [1.5, -0.5], [-0.0, 3]]
    got = oracle_lib.sample_texture(tex, uv, ws | (wt << 2) | (16 if nearest else 0))
    h, w = tex.shape[:2]

    def wrap(i, n, mode):
        if mode == 0:
            return np.mod(i, n)
        if mode == 1:
            return np.clip(i, 0, n - 1)
        m = np.mod(i, 2 * n)
        return np.where(m < n, m, 2 * n - 1 - m)
    t = tex.astype(np.float64) / 255.0
    x, y = uv[:, 0].astype(np.float64) * w, uv[:, 1].astype(np.float64) * h
    if nearest:
        want = t[wrap(np.floor(y).astype(np.int64), h, wt), wrap(np.floor(x).astype(np.int64), w, ws)]
        # a coordinate within rounding of a texel boundary may fall either side
        ok = np.isclose(got, want, atol=1e-6).all(1) | (np.abs(x - np.round(x)) < 1e-4) | (np.abs(y - np.round(y)) < 1e-4)
        assert ok.all()
        return
    x0, y0 = np.floor(x - 0.5).astype(np.int64), np.floor(y - 0.5).astype(np.int64)
    fx, fy = (x - 0.5 - x0)[:, None], (y - 0.5 - y0)[:, None]
    X0, X1, Y0, Y1 = wrap(x0, w, ws), wrap(x0 + 1, w, ws), wrap(y0, h, wt), wrap(y0 + 1, h, wt)
    want = (t[Y0, X0] * (1 - fx) + t[Y0, X1] * fx) * (1 - fy) + (t[Y1, X0] * (1 - fx) + t[Y1, X1] * fx) * fy
    assert np.abs(got - want).max() < 2e-5
