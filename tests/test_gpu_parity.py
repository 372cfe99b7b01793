"""Parity of the HIP backend (through the C ABI) with the CPU oracle, on a real MI355X.

Bar: bit-exact.  Integer results (primary-hit ids, ray counts) must be equal, and because both sides
evaluate every floating-point result with the same IEEE operation sequence (see pathtracer_amd/csrc/
pt_math.h and oracle/det_math.h), radiance, accumulation and BSDF values are compared as raw bits too:
the stated fp32 tolerance is 0 ulp.
"""
import os

import numpy as np
import pytest

from pathtracer_amd import camera, procedural
from pathtracer_amd.backend import BackendError
from pathtracer_amd.scene import Params, SceneData, make_material
from tests import oracle_lib
from tests.conftest import GOLDEN, random_frames, random_rays, random_surfaces

pytestmark = pytest.mark.gpu

M32 = 0xFFFFFFFF


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def render_pair(backend, scene, cam, W, H, frames, upload=True, **params):
    p = Params.reference(**params)
    if upload:
        backend.scene_upload(scene)
        backend.build_accel()
    backend.resize(W, H)
    backend.set_camera(cam)
    backend.set_params(p)
    o = oracle_lib.Oracle(scene)
    o.resize(W, H)
    o.set_camera(cam)
    o.set_params(p)
    r0 = backend.ray_count()
    for f in frames:
        backend.trace(f)
        o.trace(f)
    return o, backend.ray_count() - r0


def assert_same_frame(backend, o, rays_gpu):
    assert np.array_equal(backend.read_primary_ids(), o.read_primary_ids())
    assert rays_gpu == o.ray_count()
    assert np.array_equal(bits(backend.read_accum()), bits(o.read_accum()))
    assert np.array_equal(bits(backend.read_image()), bits(o.read_image()))


# ---- function level -----------------------------------------------------------------------------------
@pytest.mark.parametrize("fn", [0, 1, 2, 3])
def test_det_math_bit_exact(backend, fn):
    rng = np.random.default_rng(fn)
    x = {0: rng.uniform(0, 2 * np.pi, 200000), 1: rng.uniform(-50, 50, 200000),
         2: np.exp(rng.uniform(-100, 88, 200000)), 3: rng.uniform(-110, 90, 200000)}[fn].astype(np.float32)
    x[:6] = [0.0, -0.0, 1.0, np.inf, -np.inf, np.nan]
    assert np.array_equal(bits(backend.debug_math(fn, x)), bits(oracle_lib.det_math(fn, x)))


def test_det_pow_bit_exact(backend):
    rng = np.random.default_rng(9)
    x = rng.uniform(0, 1, 200000).astype(np.float32)
    y = rng.uniform(0, 3, 200000).astype(np.float32)
    x[:3] = [0.0, 1.0, 1e-40]
    assert np.array_equal(bits(backend.debug_math(4, x, y)), bits(oracle_lib.det_math(4, x, y)))


@pytest.mark.parametrize("kind", ["diffuse", "metal", "mixed", "full", "glass"])
def test_disney_sample_and_eval_bit_exact(backend, kind):
    rng = np.random.default_rng(hash(kind) & 0xFFFF)
    n = 20000
    s = random_surfaces(rng, n, kind)
    V, N = random_frames(rng, n)
    seeds = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    fg, Lg, pg, sg = backend.debug_disney_sample(s, V, N, seeds)
    fc, Lc, pc, sc = oracle_lib.disney_sample(s, V, N, seeds)
    assert np.array_equal(sg, sc)
    assert np.array_equal(bits(Lg), bits(Lc))
    assert np.array_equal(bits(fg), bits(fc))
    assert np.array_equal(bits(pg), bits(pc))
    L = rng.normal(size=(n, 3)).astype(np.float32)
    L /= np.linalg.norm(L, axis=1, keepdims=True)
    fg, pg = backend.debug_disney_eval(s, V, N, L)
    fc, pc = oracle_lib.disney_eval(s, V, N, L)
    assert np.array_equal(bits(fg), bits(fc)) and np.array_equal(bits(pg), bits(pc))


def test_disney_degenerate_inputs_bit_exact(backend):
    # black albedo, eta = 1, roughness 0, grazing and back-facing view vectors, normal at the pole
    s = np.zeros((6, 17), np.float32)
    s[:, 12] = 1.5
    s[:, 13:15] = 0.001
    s[:, 15] = 1.5
    s[1, 0:3] = 1.0
    s[1, 15] = 1.0
    s[1, 12] = 1.0
    s[2, 3] = 1.0
    s[3, 0:3] = 0.5
    s[3, 11] = 1.0
    s[3, 15] = 1 / 1.5
    s[4, 0:3] = 0.7
    s[4, 9] = 1.0
    s[4, 10] = 1.0
    s[5, 0:3] = 0.7
    s[5, 9] = 1.0
    N = np.tile(np.array([[0, 0, 1]], np.float32), (6, 1))
    V = np.array([[0, 0, 1], [1, 0, 0], [0, 0, 1], [0.6, 0, 0.8], [0, 1, 1e-4], [0, 0, -1]], np.float32)
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    seeds = np.arange(6, dtype=np.uint32) * 977 + 1
    g = backend.debug_disney_sample(s, V, N, seeds)
    c = oracle_lib.disney_sample(s, V, N, seeds)
    for a, b in zip(g, c):
        assert np.array_equal(bits(a), bits(b))


# ---- traversal ---------------------------------------------------------------------------------------------
def test_intersect_equals_brute_force_on_cornell(backend, cornell):
    backend.scene_upload(cornell)
    st = backend.build_accel()
    assert st.num_triangles == cornell.num_triangles == 63978
    rng = np.random.default_rng(21)
    rays = random_rays(rng, 4000, [-0.3, 0.0, -0.58], [0.25, 0.53, 0.6])
    o = oracle_lib.Oracle(cornell)
    hg, counters = backend.debug_intersect(rays)
    assert np.array_equal(hg, o.intersect(rays, use_bvh=False))
    assert counters[0] > 0 and counters[1] > 0


def test_intersect_large_batch_and_short_rays(backend, cornell):
    backend.scene_upload(cornell)
    backend.build_accel()
    rng = np.random.default_rng(22)
    o = oracle_lib.Oracle(cornell)
    rays = random_rays(rng, 300000, [-0.35, -0.05, -0.6], [0.3, 0.6, 0.8])
    rays[::3, 7] = rng.uniform(0.01, 0.5, len(rays[::3]))       # finite tmax
    rays[1::5, 3] = rng.uniform(0.01, 0.3, len(rays[1::5]))     # large tmin
    hg, _ = backend.debug_intersect(rays)
    assert np.array_equal(hg, o.intersect(rays, use_bvh=True))


def test_intersect_degenerate_rays(backend, cornell):
    backend.scene_upload(cornell)
    backend.build_accel()
    o = oracle_lib.Oracle(cornell)
    rays = np.array([
        [0, 0.25, 0.5, 1e-5, 0, 0, -1, 1e27],          # axis aligned
        [0, 0.25, 0.5, 1e-5, 0, 0, 0, 1e27],           # zero direction
        [0, 0.25, 0.5, 1e-5, np.nan, 0, -1, 1e27],     # NaN direction
        [np.nan, 0.25, 0.5, 1e-5, 0, 0, -1, 1e27],     # NaN origin
        [0, 0.25, 0.5, 1e-5, 0, 1, 0, 1e27],
        [0, 0.25, 0.5, 1e-5, 1, 0, 0, 1e27],
        [0, 0.25, 50.0, 1e-5, 0, 0, -1, 1e27],         # far outside
        [0, 0.25, 0.5, 1e-5, 0, 0, 1, 1e27],           # leaves through the open front
    ], np.float32)
    hg, _ = backend.debug_intersect(rays)
    hc = o.intersect(rays, use_bvh=False)
    assert np.array_equal(hg, hc)
    assert tuple(hg[1, 3:5]) == (M32, M32) and tuple(hg[2, 3:5]) == (M32, M32) and tuple(hg[7, 3:5]) == (M32, M32)


def test_rays_almost_parallel_to_a_coordinate_plane(backend):
    """Directions with one component of 1e-5 ... 1e-30 (or zero): 1 / d of that axis is huge.  Hits equal the brute-force
    loop, and such rays must not cost more than others: the widening of the slab test by the rounding of o / d is per
    axis, so the huge term of one axis does not open the slabs of the other two."""
    sc = procedural.sponza_class(detail=0.05)
    backend.scene_upload(sc)
    backend.build_accel()
    rng = np.random.default_rng(5)
    n = 6000
    plain = random_rays(rng, n, [-14, 0.3, -8], [14, 10.5, 8])
    flat = plain.copy()
    axis = rng.integers(0, 3, n)
    tiny = (10.0 ** rng.uniform(-30, -5, n) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    tiny[::7] = 0.0
    tiny[1::7] = -0.0
    flat[np.arange(n), 4 + axis] = tiny
    d = flat[:, 4:7].astype(np.float64)
    flat[:, 4:7] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    flat[np.arange(n), 4 + axis] = tiny
    o = oracle_lib.Oracle(sc)
    h_plain, c_plain = backend.debug_intersect(plain)
    h_flat, c_flat = backend.debug_intersect(flat)
    assert np.array_equal(h_flat, o.intersect(flat, use_bvh=False))
    assert np.array_equal(h_plain, o.intersect(plain, use_bvh=False))
    assert c_flat[0] < 1.5 * c_plain[0] and c_flat[1] < 1.5 * c_plain[1]


def test_tie_rule_and_watertight_edges(backend):
    v = np.zeros((8, 12), np.float32)
    quad = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)]
    v[0:4, 0:3] = quad
    v[4:8, 0:3] = quad
    v[:, 5:8] = (0, 0, 1)
    idx = np.array([0, 1, 2, 0, 2, 3, 0, 1, 2, 0, 2, 3], np.uint32)
    ident = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (2, 1))
    sc = SceneData(v, idx, [(0, 4, 6, 0), (0, 0, 0, 0)], [6, 6], ident, [make_material()])
    backend.scene_upload(sc)
    st = backend.build_accel()
    assert st.num_triangles == 4 and st.num_nodes == 1 and st.num_leaves == 1
    o = oracle_lib.Oracle(sc)
    rng = np.random.default_rng(1)
    rays = random_rays(rng, 2000, [-1.2, -1.2, 0], [1.2, 1.2, 1])
    rays[:, 4:7] = (0, 0, -1)
    rays[:200, 0] = rays[:200, 1]                     # exactly on the shared diagonal
    rays[200:300, 0:2] = rng.integers(-1, 2, (100, 2))  # through corners / edge midpoints
    hg, _ = backend.debug_intersect(rays)
    assert np.array_equal(hg, o.intersect(rays, use_bvh=False))
    inside = (np.abs(rays[:, 0]) <= 1) & (np.abs(rays[:, 1]) <= 1)
    assert (hg[inside, 3] == 0).all()                 # coincident submeshes: the lower submesh index wins
    assert (hg[~inside, 3] == M32).all()


def decode_child_boxes(node_words):
    """(lo, hi) arrays of shape (8, 3) and the 8 child references of one 128-byte node."""
    origin = node_words[0:3].view(np.float32).astype(np.float64)
    exps = int(node_words[3])
    step = np.array([2.0 ** (((exps >> (8 * a)) & 255) - 127) for a in range(3)])
    q = node_words[12:24]
    lo = np.zeros((8, 3))
    hi = np.zeros((8, 3))
    for k in range(8):
        for a in range(3):
            ql = (int(q[a * 2 + (k >> 2)]) >> ((k & 3) * 8)) & 255
            qh = (int(q[(3 + a) * 2 + (k >> 2)]) >> ((k & 3) * 8)) & 255
            lo[k, a] = origin[a] + ql * step[a]
            hi[k, a] = origin[a] + qh * step[a]
    return lo, hi, node_words[4:12]


@pytest.mark.parametrize("flags", [2, 0, 18, 10])
def test_hierarchy_is_well_formed(backend, cornell, flags):
    """flags 2: every triangle entered once.  flags 0: triangles with mostly empty boxes are entered once per cell they
    pass through; such a reference's boxes bound its piece only, so containment is checked for the whole triangles.
    18: no treelet restructuring either; 10: the clustered tree."""
    backend.scene_upload(cornell)
    st = backend.build_accel(flags)
    nodes, tris = backend.debug_read_bvh()
    assert nodes.shape == (st.num_nodes, 32) and tris.shape == (st.num_references, 12)
    assert st.num_references >= st.num_triangles and (not (flags & 2) or st.num_references == st.num_triangles)
    ids = tris[:, 3].astype(np.uint64) << np.uint64(32) | tris[:, 7].astype(np.uint64)
    uniq, inverse, counts = np.unique(ids, return_inverse=True, return_counts=True)
    assert len(uniq) == st.num_triangles         # every (submesh, prim) appears, once unless it was split
    whole = counts[inverse] == 1
    seen = np.zeros(st.num_references, bool)
    visited = np.zeros(st.num_nodes, bool)
    tv = tris.view(np.float32).reshape(-1, 3, 4)[:, :, :3].astype(np.float64)
    leaves = 0
    stack = [(0, [])]
    while stack:
        i, boxes = stack.pop()
        assert not visited[i]
        visited[i] = True
        lo, hi, child = decode_child_boxes(nodes[i])
        for k in range(8):
            ch = int(child[k])
            if ch == M32:
                assert (lo[k] > hi[k]).all()          # an empty child can never be entered
                continue
            mine = boxes + [(lo[k], hi[k])]
            if ch & 0x80000000:
                first, cnt = ch & 0x0FFFFFFF, ((ch >> 28) & 7) + 1
                assert cnt <= 4 and not seen[first:first + cnt].any()
                seen[first:first + cnt] = True
                assert (np.diff(ids[first:first + cnt].astype(np.int64)) >= 0).all()     # a leaf's records in (submesh, prim) order
                leaves += 1
                p = tv[first:first + cnt][whole[first:first + cnt]].reshape(-1, 3)
                for (blo, bhi) in mine:               # every ancestor's decoded box contains the triangles
                    assert (p >= blo).all() and (p <= bhi).all()
            else:
                stack.append((ch, mine))
    assert seen.all() and visited.all() and leaves == st.num_leaves


@pytest.mark.parametrize("flags", [0, 8, 9, 4, 12, 16, 24, 25])
def test_both_builders_give_the_same_hits(backend, cornell, flags):
    # 0 = Morton radix tree with treelet restructuring, 8 = clustering, 9 = the cheaper of the two (each with the
    # surface-area-optimal collapse), 4 / 12 = radix tree / clustering with the greedy collapse, + 16 = without the treelet
    # passes: the hierarchy must not change any result
    backend.scene_upload(cornell)
    backend.build_accel(flags)
    rng = np.random.default_rng(31 + flags)
    rays = random_rays(rng, 50000, [-0.35, -0.05, -0.6], [0.3, 0.6, 0.8])
    o = oracle_lib.Oracle(cornell)
    hg, _ = backend.debug_intersect(rays)
    assert np.array_equal(hg, o.intersect(rays, use_bvh=True))


def test_optimal_collapse_fills_the_nodes(backend):
    """The surface-area-optimal choice of each wide node's children against the greedy one on the same binary tree:
    fewer and fuller nodes, lower cost, leaves of at most four triangles, the same hits."""
    sc = procedural.sponza_class(detail=0.2)
    backend.scene_upload(sc)
    rng = np.random.default_rng(3)
    rays = random_rays(rng, 20000, [-14, 0.3, -8], [14, 10.5, 8])
    shapes = {}
    for flags in (4, 0):
        st = backend.build_accel(flags)
        nodes, tris = backend.debug_read_bvh()
        child = nodes[:, 4:12]
        empty = child == M32
        leaf = ((child >> 31) == 1) & ~empty
        hits, counters = backend.debug_intersect(rays)
        shapes[flags] = (st.num_nodes, st.sah_cost, (~empty).sum(1).mean(), hits, counters)
        assert ((((child[leaf] >> 28) & 7) + 1) <= 4).all()
        assert int((((child[leaf] >> 28) & 7) + 1).sum()) == st.num_references == len(tris)
    greedy, optimal = shapes[4], shapes[0]
    assert np.array_equal(greedy[3], optimal[3])
    assert optimal[0] < 0.9 * greedy[0] and optimal[1] < 0.92 * greedy[1] and optimal[2] > greedy[2] + 1.5
    assert optimal[4][1] < 0.85 * greedy[4][1]           # triangle tests for the same rays


def test_treelet_passes_lower_the_cost(backend):
    """Treelet restructuring of the Morton radix tree: a cheaper hierarchy (surface-area cost, node visits of the same rays),
    every reference still in exactly one leaf, the same hits; a clustered tree gains from the passes too."""
    sc = procedural.sponza_class(detail=0.2)
    backend.scene_upload(sc)
    rng = np.random.default_rng(4)
    rays = random_rays(rng, 20000, [-14, 0.3, -8], [14, 10.5, 8])
    out = {}
    for flags in (16, 0, 24, 8):
        st = backend.build_accel(flags)
        nodes, tris = backend.debug_read_bvh()
        child = nodes[:, 4:12]
        leaf = ((child >> 31) == 1) & (child != M32)
        assert int((((child[leaf] >> 28) & 7) + 1).sum()) == st.num_references == len(tris)
        first = np.sort(child[leaf] & 0x0FFFFFFF)
        assert len(np.unique(first)) == len(first)
        hits, counters = backend.debug_intersect(rays)
        out[flags] = (st.sah_cost, hits, counters)
    assert np.array_equal(out[0][1], out[16][1]) and np.array_equal(out[8][1], out[16][1])
    assert out[0][0] < 0.9 * out[16][0] and out[0][2][0] < 0.95 * out[16][2][0]
    assert out[8][0] < out[24][0]


# ---- frames --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("W,H,S,B,frames", [(128, 128, 1, 4, (1, 2, 3)), (96, 64, 5, 20, (1, 2)), (24, 16, 2, 3, (7,))])
def test_cornell_frames_bit_exact(backend, cornell, cameras, W, H, S, B, frames):
    o, rays = render_pair(backend, cornell, cameras["%dx%d" % (W, H)], W, H, frames, spp_per_frame=S, max_bounces=B)
    assert_same_frame(backend, o, rays)


@pytest.mark.parametrize("chains", [1, 2, 8, 16])
def test_image_does_not_depend_on_the_chain_count(backend, cornell, cameras, chains):
    """Frames issued back to back (no synchronisation between them) with the frame split into 1, 2, 8 or 16
    independent pipelines: same bits as the oracle, hence as each other.  72x40 leaves some shards of the
    ray queue without a tile, 8 and 16 chains then have empty pipelines."""
    try:
        backend.set_chains(chains)
        for W, H in ((72, 40), (256, 256)):
            o, rays = render_pair(backend, cornell, cameras["512x512"], W, H, (1, 2, 3, 4, 5), spp_per_frame=2, max_bounces=5)
            assert_same_frame(backend, o, rays)
        t_before = backend.ray_count()
        backend.set_profiling(1)
        backend.trace(6)
        t = backend.last_trace_timing()
        backend.set_profiling(0)
        assert t.chains == chains and t.rays == backend.ray_count() - t_before
        assert 0 < t.extend_busy_ms <= t.extend_ms * 1.001 and t.extend_busy_ms <= t.total_ms * 1.001
    finally:
        backend.set_chains(4)
    with pytest.raises(BackendError):
        backend.set_chains(3)


@pytest.mark.parametrize("batch", [1, 2, 3, 4, 8])
def test_image_does_not_depend_on_frame_batching(backend, cornell, cameras, batch):
    """Up to eight consecutive trace() calls share their launches.  Ten frames issued with no synchronisation in
    between, the camera and (once) the parameters changing from frame to frame, so that short batches are launched
    while full ones are still running: same bits as the oracle tracing them one by one."""
    def moved(cam, k):
        cam = np.array(cam, np.float32).reshape(5, 4, 4).copy()
        cam[3, 3, 0] += 0.01 * k          # InverseView, column-major: translation x
        cam[3, 3, 1] -= 0.004 * k
        return cam.reshape(-1)
    try:
        backend.set_batch(batch)
        for W, H in ((128, 128), (200, 120)):
            backend.scene_upload(cornell)
            backend.build_accel()
            backend.resize(W, H)
            p = Params.reference(spp_per_frame=2, max_bounces=5)
            backend.set_params(p)
            r0 = backend.ray_count()
            calls = []
            for n, f in enumerate((1, 2, 3, 1, 2, 3, 4, 5, 6, 7)):     # the fourth call restarts the accumulation inside a batch
                if n == 5:
                    p = Params.reference(spp_per_frame=1, max_bounces=3)
                    backend.set_params(p)
                cam = moved(cameras["512x512"], n)
                backend.set_camera(cam)
                backend.trace(f)
                calls.append((f, cam, p))
            got_accum, got_image, got_ids = backend.read_accum(), backend.read_image(), backend.read_primary_ids()
            rays = backend.ray_count() - r0
            o = oracle_lib.Oracle(cornell)
            o.resize(W, H)
            for f, cam, q in calls:
                o.set_camera(cam)
                o.set_params(q)
                o.trace(f)
            assert rays == o.ray_count()
            assert np.array_equal(got_ids, o.read_primary_ids())
            assert np.array_equal(bits(got_accum), bits(o.read_accum()))
            assert np.array_equal(bits(got_image), bits(o.read_image()))
    finally:
        backend.set_batch(8)
    with pytest.raises(BackendError):
        backend.set_batch(9)


@pytest.mark.parametrize("defer", ["0", "1", "4"])
def test_image_does_not_depend_on_ray_suspension(cornell, cameras, monkeypatch, defer):
    """PT_DEFER = rounds a path may fall behind because the tail of a traversal launch suspended its ray (default 2;
    0 = never suspend).  A tiny image on many waves makes nearly every launch all tail."""
    from pathtracer_amd.backend import Backend
    monkeypatch.setenv("PT_DEFER", defer)
    b = Backend(0)
    try:
        for W, H, S, B in ((64, 48, 3, 6), (256, 200, 1, 8)):
            o, rays = render_pair(b, cornell, cameras["512x512"], W, H, (1, 2, 3), spp_per_frame=S, max_bounces=B)
            assert_same_frame(b, o, rays)
    finally:
        b.close()


def test_iteration_bound_of_the_traversal_is_reported(cornell, cameras, monkeypatch):
    """A traversal wave leaves its loop after a fixed number of iterations whatever the data (2^24 by default, which no
    well-formed hierarchy reaches).  With the bound forced down to 3 (PT_ITER_CAP) every wave reaches it and leaves hit records
    unwritten: the shading stage survives whatever those records hold (a triangle number beyond the table is a miss), and the next
    synchronising call returns PT_ERR_TRAVERSAL instead of handing out the frame silently.  The context stays usable."""
    from pathtracer_amd.backend import Backend
    monkeypatch.setenv("PT_ITER_CAP", "3")
    b = Backend(0)
    try:
        b.scene_upload(cornell)
        b.build_accel()
        b.resize(64, 64)
        b.set_camera(cameras["64x64"])
        b.set_params(Params.reference(spp_per_frame=1, max_bounces=4))
        b.trace(1)
        with pytest.raises(BackendError) as e:
            b.sync()
        assert e.value.code == 6 and "iteration bound" in str(e.value)
        b.sync()                                 # the fault was reported once
        rays = random_rays(np.random.default_rng(3), 1000, -0.3, 0.5)
        with pytest.raises(BackendError) as e:
            b.debug_intersect(rays)
        assert e.value.code == 6
    finally:
        b.close()
    monkeypatch.delenv("PT_ITER_CAP")
    b = Backend(0)
    try:
        o, rays = render_pair(b, cornell, cameras["64x64"], 64, 64, (1, 2), spp_per_frame=1, max_bounces=4)
        assert_same_frame(b, o, rays)
    finally:
        b.close()


def test_cornell_512_config_and_golden_ids(backend, cornell, cameras):
    # the CPU-runnable configuration: Cornell box 512x512, 1 spp, 4 bounces
    o, rays = render_pair(backend, cornell, cameras["512x512"], 512, 512, (1, 2), spp_per_frame=1, max_bounces=4)
    assert_same_frame(backend, o, rays)
    # 64x64 ids against the committed brute-force fixture
    backend.resize(64, 64)
    backend.set_camera(cameras["64x64"])
    backend.trace(1)
    assert np.array_equal(backend.read_primary_ids(), np.load(os.path.join(GOLDEN, "cornell_primary_ids_64.npy")))


def test_cornell_golden_radiance(backend, cornell, cameras):
    backend.scene_upload(cornell)
    backend.build_accel()
    backend.resize(128, 128)
    backend.set_camera(cameras["128x128"])
    backend.set_params(Params.reference(spp_per_frame=1, max_bounces=4))
    for f in range(1, 18):
        backend.trace(f)
    want = np.load(os.path.join(GOLDEN, "cornell_radiance_128_16spp.npy"))
    assert np.array_equal(bits(backend.read_accum()), bits(want))


def test_ragged_sizes(backend, cornell):
    backend.scene_upload(cornell)
    backend.build_accel()
    for (W, H) in [(1, 1), (7, 3), (9, 17), (65, 8), (8, 65)]:
        cam = camera.cornell_camera(W, H)
        o, rays = render_pair(backend, cornell, cam, W, H, (1, 2), upload=False, spp_per_frame=2, max_bounces=3)
        assert_same_frame(backend, o, rays)


def test_metal_variant_and_material_update(backend, cornell, cameras):
    # the "metal" variant of the Cornell config: sphere material -> metallic 1, roughness 0.2
    mats = cornell.materials.copy()
    mats[1] = make_material(albedo=(0.9, 0.6, 0.3), metallic=1.0, roughness=0.2)
    sc = SceneData(cornell.vertices, cornell.indices, cornell.submeshes, cornell.index_counts, cornell.xforms, mats)
    o, rays = render_pair(backend, sc, cameras["96x64"], 96, 64, (1, 2), spp_per_frame=1, max_bounces=8)
    assert_same_frame(backend, o, rays)
    # UpdateMaterialData (RayTracingLayer.cpp:621): new materials, same geometry and hierarchy
    mats2 = mats.copy()
    mats2[0] = make_material(albedo=(0.3, 0.3, 0.8), roughness=0.4, clearcoat=1.0, clearcoat_roughness=0.1)
    backend.update_materials(mats2)
    o.update_materials(mats2)
    r0, c0 = backend.ray_count(), o.ray_count()
    for f in (1, 2, 3):
        backend.trace(f)
        o.trace(f)
    assert backend.ray_count() - r0 == o.ray_count() - c0
    assert np.array_equal(bits(backend.read_accum()), bits(o.read_accum()))


def test_full_disney_procedural_scene(backend):
    # small instance of the Sponza-class stand-in: clearcoat / sheen / transmission materials, scaled submeshes
    sc = procedural.sponza_class(detail=0.05)
    W, H = 160, 90
    o, rays = render_pair(backend, sc, procedural.sponza_camera(W, H), W, H, (1, 2, 3), spp_per_frame=1, max_bounces=8)
    assert_same_frame(backend, o, rays)
    img = backend.read_image()
    assert len(np.unique(backend.read_primary_ids()[..., 0])) > 20 and np.isfinite(img).all()


@pytest.mark.parametrize("workload", ["sponza_like", "sponza_class"])
def test_benchmark_workload_at_full_size(backend, workload):
    """The configurations bench.py times at 1920x1080, 8 bounces, frames issued back to back in one batch: the headline
    (the textured Sponza-like atrium with split triangles under the reference's Preetham sky as a 2048-texel cube, mirrored below
    the horizon where that shader has no values, so that every miss is a real environment lookup) and the uniformly tessellated Sponza-class scene under the constant sky.  Ray
    count, primary ids and every pixel's bits equal the oracle's (24-26 M ray segments, a few seconds of the host's cores)."""
    import copy
    W, H = 1920, 1080
    if workload == "sponza_like":
        sc = procedural.sponza_class(layout="sponza", textures=True)
        sky = (3.14, 0.0, 0.0, 2048)
        backend.scene_upload(sc)
        st = backend.build_accel()
        assert st.num_references > st.num_triangles          # large sparse triangles were split
        backend.env_preetham(*sky)
        backend.env_upload(procedural.sky_mirrored_below_the_horizon(backend.read_env()))      # as bench.py does: finite below the horizon
        sco = copy.copy(sc)
        sco.env = procedural.sky_mirrored_below_the_horizon(oracle_lib.preetham_sky(*sky))
        p = Params.reference(spp_per_frame=1, max_bounces=8)
        backend.resize(W, H)
        backend.set_camera(procedural.sponza_camera(W, H))
        backend.set_params(p)
        o = oracle_lib.Oracle(sco)
        o.resize(W, H)
        o.set_camera(procedural.sponza_camera(W, H))
        o.set_params(p)
        r0 = backend.ray_count()
        for f in (1, 2, 3):
            backend.trace(f)
            o.trace(f)
        rays = backend.ray_count() - r0
        ids = backend.read_primary_ids()
        assert (ids[..., 0] == M32).sum() > 1000              # the sky is in view
        img = backend.read_image()
        assert np.isfinite(img).all() and not ((img[..., 0] == 1.0) & (img[..., 1] == 0.0) & (img[..., 2] == 0.0)).any()
    else:
        sc = procedural.sponza_class()
        o, rays = render_pair(backend, sc, procedural.sponza_camera(W, H), W, H, (1, 2, 3), spp_per_frame=1, max_bounces=8)
    assert rays > 30_000_000
    assert_same_frame(backend, o, rays)
    o.close()


def test_long_progressive_run_keeps_its_state(backend):
    """4096 samples per pixel, the progressive configuration: frames 1..4090 on the GPU alone (over 500 batches of
    launches through the queues, the suspension records and the counters), then the accumulation image goes to the oracle and
    frames 4091..4093 run on both: every bit equal, so nothing drifted on the way.  1920x1080, Sponza-class, 8 bounces."""
    sc = procedural.sponza_class()
    W, H, N = 1920, 1080, 4090
    cam = procedural.sponza_camera(W, H)
    p = Params.reference(spp_per_frame=1, max_bounces=8)
    backend.scene_upload(sc)
    backend.build_accel()
    backend.resize(W, H)
    backend.set_camera(cam)
    backend.set_params(p)
    counts = [backend.ray_count()]
    for f in range(1, N + 1):
        backend.trace(f)
        if f % 1024 == 0:
            counts.append(backend.ray_count())
    counts.append(backend.ray_count())
    steps = np.diff(counts)
    assert (steps[:-1] > 0).all() and abs(steps[1] - steps[2]) < 0.01 * steps[1]      # the same work per thousand frames
    acc = backend.read_accum()
    img = backend.read_image()
    assert (acc[..., 3] == N - 1).all()                       # frame 1 clears, every later frame adds one sample
    assert np.isfinite(acc).all() and np.isfinite(img).all()
    assert not ((img[..., 0] == 1.0) & (img[..., 1] == 0.0) & (img[..., 2] == 0.0)).any()     # no NaN pixel was painted red
    o = oracle_lib.Oracle(sc)
    o.resize(W, H)
    o.set_camera(cam)
    o.set_params(p)
    o.write_accum(acc)
    r0 = backend.ray_count()
    for f in (N + 1, N + 2, N + 3):
        backend.trace(f)
        o.trace(f)
    assert backend.ray_count() - r0 == o.ray_count()
    assert np.array_equal(bits(backend.read_accum()), bits(o.read_accum()))
    assert np.array_equal(bits(backend.read_image()), bits(o.read_image()))
    assert np.array_equal(backend.read_primary_ids(), o.read_primary_ids())
    o.close()


@pytest.mark.parametrize("batch", [1, 2, 8])
def test_suspension_follows_the_batch_size(backend, batch):
    """Unless PT_DEFER forces it, a batch of one or two frames over a deep hierarchy (surface-area cost >= 20) suspends the last
    rays of its traversal launches and takes two more rounds, larger batches do not.  Seven frames issued back to back with
    1, 2 or 8 frames per batch, then one more frame after a flush (a batch of one following whatever came before on the same
    chains and counter blocks): same bits as the oracle, whichever kernel variants ran."""
    sc = procedural.sponza_class(detail=0.2)
    W, H = 240, 135
    try:
        backend.set_batch(batch)
        o, rays = render_pair(backend, sc, procedural.sponza_camera(W, H), W, H, (1, 2, 3, 4, 5, 6, 7), spp_per_frame=1, max_bounces=6)
        assert backend.accel.sah_cost >= 20.0           # the rule applies to this scene
        assert_same_frame(backend, o, rays)
        r0 = backend.ray_count()
        backend.trace(8)
        o.trace(8)
        assert np.array_equal(bits(backend.read_accum()), bits(o.read_accum()))
        assert rays + backend.ray_count() - r0 == o.ray_count()
    finally:
        backend.set_batch(8)


def test_large_scene(backend):
    """The Bistro-class street: 2.97 M triangles, an 11-level hierarchy that outgrows every cache, long grazing rays
    (the longest traversals spill their stacks).  Ray count, primary ids and every pixel's bits equal the oracle's."""
    sc = procedural.bistro_class()
    W, H = 320, 180
    o, rays = render_pair(backend, sc, procedural.bistro_camera(W, H), W, H, (1, 2, 3), spp_per_frame=1, max_bounces=8)
    assert backend.accel.num_triangles > 2_900_000 and backend.accel.max_depth >= 10
    assert_same_frame(backend, o, rays)


def _world_triangles(sc):
    """(T, 3, 3) float64 world-space corners of every triangle, in (submesh, prim) order."""
    out = []
    for (first_index, first_vertex, _, _), count, xf in zip(sc.submeshes, sc.index_counts, sc.xforms):
        idx = sc.indices[int(first_index):int(first_index) + int(count)].astype(np.int64) + int(first_vertex)
        m = np.asarray(xf, np.float64).reshape(3, 4)
        p = sc.vertices[idx, 0:3].astype(np.float64) @ m[:, :3].T + m[:, 3]
        out.append(p.reshape(-1, 3, 3))
    return np.concatenate(out)


def test_split_triangles_change_no_result(backend):
    """Architectural geometry: wall / floor quads spanning the scene and long thin diagonal beams.  The builder enters
    such triangles once per cell they pass through.  Hits must not change: rays aimed at points spread over the split
    triangles (and random rays) equal the oracle's brute-force loop and the hierarchy built without splitting, word for
    word; every reference still names an original (submesh, prim)."""
    sc = procedural.sponza_class(detail=0.05, layout="sponza", textures=True)
    backend.scene_upload(sc)
    whole = backend.build_accel(2)
    assert whole.num_references == whole.num_triangles == sc.num_triangles
    rng = np.random.default_rng(77)
    tri = _world_triangles(sc)
    e = tri.max(1) - tri.min(1)
    box_area = 2 * (e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0])
    big = np.argsort(-box_area)[:400]
    pick = rng.choice(big, 6000)
    w = rng.dirichlet((1, 1, 1), len(pick))
    target = (tri[pick] * w[:, :, None]).sum(1)
    origin = rng.uniform([-14, 0.3, -8], [14, 10.5, 8], (len(pick), 3))
    d = target - origin
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    aimed = np.zeros((len(pick), 8), np.float32)
    aimed[:, 0:3], aimed[:, 3], aimed[:, 4:7], aimed[:, 7] = origin, 1e-4, d, 1e27
    rays = np.concatenate([aimed, random_rays(rng, 6000, [-14, 0.3, -8], [14, 10.5, 8])])
    h_whole, _ = backend.debug_intersect(rays)
    st = backend.build_accel(0)
    assert st.num_triangles == sc.num_triangles and st.num_references > st.num_triangles
    assert st.sah_cost < whole.sah_cost
    h_split, _ = backend.debug_intersect(rays)
    assert np.array_equal(h_split, h_whole)
    o = oracle_lib.Oracle(sc)
    assert np.array_equal(h_split, o.intersect(rays, use_bvh=False))
    assert (h_split[:len(pick), 3] != M32).mean() > 0.95
    nodes, tris = backend.debug_read_bvh()
    assert tris.shape == (st.num_references, 12)
    ids = tris[:, 3].astype(np.uint64) << np.uint64(32) | tris[:, 7].astype(np.uint64)
    uniq, counts = np.unique(ids, return_counts=True)
    assert len(uniq) == st.num_triangles and counts.max() <= 64 and (counts > 1).sum() > 50
    # all references of a triangle carry the same vertices and ids; word 11 is 0 for a triangle entered whole, and for a
    # reference to a part the bounds of the part's weights: (origin vertex + 1) << 28 | u0 | v0 << 7 | u1 << 14 | v1 << 21
    order = np.argsort(ids, kind="stable")
    same = ids[order][1:] == ids[order][:-1]
    assert np.array_equal(tris[order][1:][same][:, :11], tris[order][:-1][same][:, :11])
    split_ids = uniq[counts > 1]
    is_part = np.isin(ids, split_ids)
    code = tris[:, 11]
    assert (code[~is_part] == 0).all() and (code[is_part] != 0).all()
    u0, v0, u1, v1 = code & 127, (code >> 7) & 127, (code >> 14) & 127, (code >> 21) & 127
    assert (u0[is_part] <= u1[is_part]).all() and (v0[is_part] <= v1[is_part]).all() and ((code[is_part] >> 28) - 1 <= 2).all()
    assert ((u1 - u0)[is_part].astype(np.int64) * (v1 - v0)[is_part] < 127 * 127 // 4).mean() > 0.95     # parts, not the whole range


def test_refit_of_split_triangles_under_new_transforms(backend):
    """A refit keeps the references of split triangles tight: each is bounded again from the barycentric bounds of its part,
    which no affine transform changes.  Unchanged transforms: the refitted hierarchy costs little more than the built one and
    less than the hierarchy without splitting.  Then a third of the submeshes is rotated, sheared and moved: hits after the
    refit equal the oracle's brute-force loop over the moved triangles."""
    import copy
    sc = procedural.sponza_class(detail=0.05, layout="sponza", textures=True)
    backend.scene_upload(sc)
    whole = backend.build_accel(2).sah_cost
    built = backend.build_accel(0)
    assert built.num_references > built.num_triangles
    built_cost = built.sah_cost
    same = backend.update_transforms(sc.xforms, flags=0x100)
    assert same.sah_cost < built_cost * 1.1 and same.sah_cost < 0.5 * (built_cost + whole), (built_cost, same.sah_cost, whole)
    rng = np.random.default_rng(5)
    xf = sc.xforms.copy().reshape(-1, 3, 4)
    for i in range(0, len(xf), 3):
        a = rng.uniform(-0.3, 0.3)
        rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        shear = np.eye(3)
        shear[0, 1] = rng.uniform(-0.2, 0.2)
        m = rot @ shear
        xf[i, :, :3] = (m @ xf[i, :, :3].astype(np.float64)).astype(np.float32)
        xf[i, :, 3] = (m @ xf[i, :, 3].astype(np.float64) + rng.uniform(-0.5, 0.5, 3)).astype(np.float32)
    moved = copy.copy(sc)
    moved.xforms = np.ascontiguousarray(xf.reshape(-1, 12))
    st = backend.update_transforms(moved.xforms, flags=0x100)
    assert st.num_references == built.num_references
    rays = random_rays(rng, 20000, [-14, 0.3, -8], [14, 10.5, 8])
    tri = _world_triangles(moved)
    e = tri.max(1) - tri.min(1)
    big = np.argsort(-(e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0]))[:400]
    pick = rng.choice(big, 6000)
    w = rng.dirichlet((1, 1, 1), len(pick))
    target = (tri[pick] * w[:, :, None]).sum(1)
    origin = rng.uniform([-14, 0.3, -8], [14, 10.5, 8], (len(pick), 3))
    d = target - origin
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    aimed = np.zeros((len(pick), 8), np.float32)
    aimed[:, 0:3], aimed[:, 3], aimed[:, 4:7], aimed[:, 7] = origin, 1e-4, d, 1e27
    rays = np.concatenate([aimed, rays])
    got, _ = backend.debug_intersect(rays)
    o = oracle_lib.Oracle(moved)
    assert np.array_equal(got, o.intersect(rays, use_bvh=False))
    assert (got[:len(pick), 3] != M32).mean() > 0.9
    o.close()


def test_split_triangles_frames_bit_exact(backend):
    """Frames of the architectural scene (textures on every material, split triangles) equal the oracle's, with the
    split hierarchy, after a refit of it, and without splitting."""
    sc = procedural.sponza_class(detail=0.05, layout="sponza", textures=True)
    W, H = 160, 90
    cam = procedural.sponza_camera(W, H)
    o, rays = render_pair(backend, sc, cam, W, H, (1, 2, 3), spp_per_frame=1, max_bounces=8)
    assert backend.accel.num_references > backend.accel.num_triangles
    assert_same_frame(backend, o, rays)
    ref = backend.read_accum()
    backend.update_transforms(sc.xforms, flags=0x100)
    r0 = backend.ray_count()
    for f in (1, 2, 3):
        backend.trace(f)
    assert backend.ray_count() - r0 == rays and np.array_equal(bits(backend.read_accum()), bits(ref))
    backend.build_accel(2)
    for f in (1, 2, 3):
        backend.trace(f)
    assert np.array_equal(bits(backend.read_accum()), bits(ref))


def test_textured_reference_asset_at_1080p_64_frames(backend):
    """BASELINE's second configuration as written: a glTF asset of the reference (Suzanne.gltf with its base-colour and
    metallic-roughness PNGs, ClosestHit.glsl:176-186), 1920x1080, 1 spp x 64 accumulated frames (frame indices 1..65),
    8 bounces.  Primary ids, ray count, the 64-frame accumulation and the final image equal the oracle's bit for bit
    (relRMSE 0); at 240x135 the primary ids also equal the oracle's brute-force loop over all triangles."""
    sc = SceneData.load(os.path.join(GOLDEN, "suzanne_scene.npz"))
    cams = np.load(os.path.join(GOLDEN, "suzanne_cameras.npz"))
    assert len(sc.textures) == 2 and sc.num_triangles == 3936
    W, H = 1920, 1080
    o, rays = render_pair(backend, sc, cams["1920x1080"], W, H, range(1, 66), spp_per_frame=1, max_bounces=8)
    assert_same_frame(backend, o, rays)
    acc = backend.read_accum()
    assert (acc[..., 3] == 64).all()
    ids = backend.read_primary_ids()
    assert 0.2 < (ids[..., 0] != M32).mean() < 0.6
    o.close()
    W, H = 240, 135
    o, rays = render_pair(backend, sc, cams["240x135"], W, H, (1,), upload=False, spp_per_frame=1, max_bounces=1)
    brute = oracle_lib.Oracle(sc)
    brute.resize(W, H)
    brute.set_camera(cams["240x135"])
    brute.set_params(Params.reference(spp_per_frame=1, max_bounces=1))
    brute.trace(1, use_bvh=False)
    assert np.array_equal(backend.read_primary_ids(), brute.read_primary_ids())
    # the textures matter: the same frames with the maps switched off give another picture
    plain = SceneData(sc.vertices, sc.indices, sc.submeshes, sc.index_counts, sc.xforms, sc.materials.copy())
    plain.materials.view(np.int32)[:, 10:13] = -1
    o2, _ = render_pair(backend, plain, cams["240x135"], W, H, (1, 2), spp_per_frame=1, max_bounces=8)
    untextured = backend.read_accum()
    o3, rays3 = render_pair(backend, sc, cams["240x135"], W, H, (1, 2), spp_per_frame=1, max_bounces=8)
    assert_same_frame(backend, o3, rays3)
    assert not np.array_equal(backend.read_accum(), untextured)


def _export_gltf(sc, folder):
    """Writes a SceneData as .gltf + .bin + PNG files (one node per submesh, its 3x4 matrix as the node matrix)."""
    import json
    from PIL import Image
    blob = bytearray()
    views, accessors, meshes, nodes = [], [], [], []

    def add(arr, ctype, atype, target=None):
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": arr.nbytes})
        blob.extend(arr.tobytes())
        acc = {"bufferView": len(views) - 1, "componentType": ctype, "count": len(arr), "type": atype}
        if atype == "VEC3" and target == "pos":
            acc["min"], acc["max"] = arr.min(0).tolist(), arr.max(0).tolist()
        accessors.append(acc)
        return len(accessors) - 1
    for i, (sub, cnt, xf) in enumerate(zip(sc.submeshes, sc.index_counts, sc.xforms)):
        vo, io, mat = int(sub[1]), int(sub[2]), int(sub[3])
        idx = sc.indices[io:io + int(cnt)]
        nv = int(idx.max()) + 1
        v = sc.vertices[vo:vo + nv]
        prim = {"attributes": {"POSITION": add(np.ascontiguousarray(v[:, 0:3]), 5126, "VEC3", "pos"), "TEXCOORD_0": add(np.ascontiguousarray(v[:, 3:5]), 5126, "VEC2"),
                               "NORMAL": add(np.ascontiguousarray(v[:, 5:8]), 5126, "VEC3"), "TANGENT": add(np.ascontiguousarray(v[:, 8:12]), 5126, "VEC4")},
                "indices": add(np.ascontiguousarray(idx, np.uint32), 5125, "SCALAR"), "material": mat}
        meshes.append({"primitives": [prim]})
        m = np.eye(4)
        m[:3, :] = xf.reshape(3, 4)
        nodes.append({"mesh": i, "name": sc.names[i], "matrix": m.T.reshape(-1).tolist()})
    mats = []
    for w in sc.materials:
        wi = w.view(np.int32)
        pbr = {"baseColorFactor": [float(w[0]), float(w[1]), float(w[2]), 1.0], "metallicFactor": float(w[3]), "roughnessFactor": float(w[4])}
        if wi[10] >= 0:
            pbr["baseColorTexture"] = {"index": int(wi[10])}
        if wi[11] >= 0:
            pbr["metallicRoughnessTexture"] = {"index": int(wi[11])}
        mats.append({"pbrMetallicRoughness": pbr, "emissiveFactor": [float(w[5]), float(w[6]), float(w[7])],
                     "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": float(w[8])}, "KHR_materials_ior": {"ior": float(w[21])}}})
    images = []
    for i, t in enumerate(sc.textures):
        Image.fromarray(t, "RGBA").save(os.path.join(folder, "tex%d.png" % i))
        images.append({"uri": "tex%d.png" % i})
    g = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": list(range(len(nodes)))}], "nodes": nodes, "meshes": meshes, "materials": mats,
         "accessors": accessors, "bufferViews": views, "buffers": [{"uri": "scene.bin", "byteLength": len(blob)}],
         "images": images, "textures": [{"source": i, "sampler": 0} for i in range(len(images))], "samplers": [{}]}
    open(os.path.join(folder, "scene.bin"), "wb").write(bytes(blob))
    path = os.path.join(folder, "scene.gltf")
    json.dump(g, open(path, "w"))
    return path


def test_bench_workload_from_a_gltf_file(backend, tmp_path):
    """bench.py --gltf: a user's own model by path (RayTracingLayer.cpp:27-32), imported with its textures by the C++ host layer.
    The textured fixture goes out as .gltf + .bin + PNG, comes back through the same code bench.py runs, and renders bit for bit
    like the fixture itself."""
    pytest.importorskip("PIL.Image")
    import bench
    sc = SceneData.load(os.path.join(GOLDEN, "suzanne_scene.npz"))
    path = _export_gltf(sc, str(tmp_path))
    W, H = 160, 90
    got, cam, text = bench.gltf_workload(path, W, H, 1.0, "0.9,0.4,2.5,0,-0.02,0")
    assert "3936 tris" in text and "2 textures" in text
    assert np.array_equal(got.vertices, sc.vertices) and np.array_equal(got.indices, sc.indices) and np.array_equal(got.xforms, sc.xforms)
    assert np.array_equal(bits(got.materials), bits(sc.materials)) and all(np.array_equal(a, b) for a, b in zip(got.textures, sc.textures))
    o, rays = render_pair(backend, got, cam, W, H, (1, 2, 3), spp_per_frame=1, max_bounces=6)
    assert_same_frame(backend, o, rays)
    assert (backend.read_primary_ids()[..., 0] != M32).mean() > 0.2
    # the default camera frames the model
    _, cam2, _ = bench.gltf_workload(path, W, H, 1.0, None)
    o, rays = render_pair(backend, got, cam2, W, H, (1,), upload=False, spp_per_frame=1, max_bounces=1)
    assert (backend.read_primary_ids()[..., 0] != M32).mean() > 0.05


def test_russian_roulette_option(backend, cornell, cameras):
    o, rays = render_pair(backend, cornell, cameras["96x64"], 96, 64, (1, 2), spp_per_frame=2, max_bounces=12, rr_depth=2)
    assert_same_frame(backend, o, rays)
    o2, rays2 = render_pair(backend, cornell, cameras["96x64"], 96, 64, (1, 2), upload=False, spp_per_frame=2, max_bounces=12)
    assert rays < rays2          # roulette ends paths early


def test_textures_and_environment_map(backend, cornell, cameras):
    rng = np.random.default_rng(5)
    tex0 = rng.integers(0, 256, (16, 32, 4), dtype=np.uint8)
    tex1 = rng.integers(0, 256, (5, 3, 4), dtype=np.uint8)
    env = rng.uniform(0, 2, (6, 4, 4, 4)).astype(np.float32)
    verts = cornell.vertices.copy()
    verts[:, 3:5] = verts[:, 0:2] * 3.7 - 0.4        # the Cornell asset has no uv set: derive one, with wrap-around
    mats = cornell.materials.copy()
    mats[0] = make_material(albedo=(1, 1, 1), roughness=1.0, albedo_map=0, mr_map=1, metallic=1.0)
    mats[1] = make_material(albedo=(1, 0.5, 0.5), roughness=0.8, albedo_map=1)
    sc = SceneData(verts, cornell.indices, cornell.submeshes, cornell.index_counts, cornell.xforms, mats,
                   textures=[tex0, tex1], env=env)
    o, rays = render_pair(backend, sc, cameras["96x64"], 96, 64, (1, 2, 3), spp_per_frame=1, max_bounces=5)
    assert_same_frame(backend, o, rays)


@pytest.mark.parametrize("flags", [1 | (1 << 2), 2 | (2 << 2), 1 | (2 << 2) | 16, 16])
def test_texture_sampler_state(backend, cornell, cameras, flags):
    """Wrap modes (clamp to edge, mirrored repeat) and the nearest filter of the scene's textures: same bits as
    the oracle, and a picture that differs from the bilinear / repeat one."""
    rng = np.random.default_rng(6)
    tex0 = rng.integers(0, 256, (16, 32, 4), dtype=np.uint8)
    tex1 = rng.integers(0, 256, (5, 3, 4), dtype=np.uint8)
    verts = cornell.vertices.copy()
    verts[:, 3:5] = verts[:, 0:2] * 3.7 - 0.4        # uv outside [0, 1] on most of the box
    mats = cornell.materials.copy()
    mats[0] = make_material(albedo=(1, 1, 1), roughness=1.0, albedo_map=0, mr_map=1, metallic=1.0)
    mats[1] = make_material(albedo=(1, 0.5, 0.5), roughness=0.8, albedo_map=1)
    sc = SceneData(verts, cornell.indices, cornell.submeshes, cornell.index_counts, cornell.xforms, mats, textures=[tex0, tex1])
    o, rays = render_pair(backend, sc, cameras["96x64"], 96, 64, (1, 2), spp_per_frame=1, max_bounces=4)
    plain = backend.read_accum()
    sc.texture_flags = [flags, flags]
    o, rays = render_pair(backend, sc, cameras["96x64"], 96, 64, (1, 2), spp_per_frame=1, max_bounces=4)
    assert_same_frame(backend, o, rays)
    assert not np.array_equal(backend.read_accum(), plain)
    bad = SceneData(verts, cornell.indices, cornell.submeshes, cornell.index_counts, cornell.xforms, mats, textures=[tex0, tex1])
    bad.texture_flags = [3, 0]
    with pytest.raises(BackendError):
        backend.scene_upload(bad)
    backend.scene_upload(cornell)
    backend.build_accel()


def test_nan_pixels_are_red_like_the_oracle(backend, cornell, cameras):
    mats = cornell.materials.copy()
    mats[0] = make_material(emissive=(0, 0, 0), emissive_strength=np.inf)
    sc = SceneData(cornell.vertices, cornell.indices, cornell.submeshes, cornell.index_counts, cornell.xforms, mats)
    o, rays = render_pair(backend, sc, cameras["24x16"], 24, 16, (1, 2), spp_per_frame=1, max_bounces=3)
    assert_same_frame(backend, o, rays)
    img = backend.read_image()
    assert (img == np.array([1, 0, 0, 1], np.float32)).all(-1).any()


def test_empty_scene_and_single_triangle(backend):
    ident = np.array([[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]], np.float32)
    empty = SceneData(np.zeros((1, 12), np.float32), np.zeros(0, np.uint32), [(0, 0, 0, 0)], [0], ident, [make_material()])
    cam = camera.cornell_camera(16, 8)
    o, rays = render_pair(backend, empty, cam, 16, 8, (1, 2), spp_per_frame=1, max_bounces=4)
    assert backend.accel.num_nodes == 0 and rays == 16 * 8 * 2
    assert_same_frame(backend, o, rays)
    v = np.zeros((3, 12), np.float32)
    v[:, 0:3] = [(-1, -1, -0.5), (1, -1, -0.5), (0, 1.5, -0.5)]
    v[:, 5:8] = (0, 0, 1)
    one = SceneData(v, [0, 1, 2], [(0, 0, 0, 0)], [3], ident, [make_material(emissive=(1, 2, 3))])
    o, rays = render_pair(backend, one, cam, 16, 8, (1, 2), spp_per_frame=1, max_bounces=4)
    assert backend.accel.num_nodes == 1
    assert_same_frame(backend, o, rays)


# ---- size-independent properties at the full 1080p size ---------------------------------------------------------
def test_full_size_accumulation_properties(backend, cornell, cameras):
    W, H = 1920, 1080
    backend.scene_upload(cornell)
    backend.build_accel()
    backend.resize(W, H)
    backend.set_camera(cameras["1920x1080"])
    backend.set_params(Params.reference(spp_per_frame=1, max_bounces=8))
    r0 = backend.ray_count()
    backend.trace(1)
    assert not backend.read_accum().any()                         # frame 1 clears (RayGen.glsl:361-365)
    n1 = backend.ray_count() - r0
    assert W * H <= n1 <= W * H * 8
    backend.trace(2)
    a2 = backend.read_accum()
    backend.trace(3)
    a3 = backend.read_accum()
    assert (a2[..., 3] == 1).all() and (a3[..., 3] == 2).all()
    assert (a3[..., :3] >= a2[..., :3]).all()                     # radiance is non-negative, sums only grow
    # idempotence: the same frame index on a restored accumulation gives the same bits
    backend.write_accum(a2)
    backend.trace(3)
    assert np.array_equal(bits(backend.read_accum()), bits(a3))
    # merge property used by the multi-GPU split: frames rendered into separate buffers add up
    backend.write_accum(np.zeros_like(a2))
    backend.trace(3)
    only3 = backend.read_accum()
    assert np.allclose(only3[..., :3] + a2[..., :3], a3[..., :3], rtol=1e-6, atol=1e-6) and (only3[..., 3] == 1).all()
    # resolve: image = accum.rgb / accum.a
    backend.write_accum(a3)
    backend.resolve()
    img = backend.read_image()
    assert np.array_equal(bits(img[..., :3]), bits((a3[..., :3] / a3[..., 3:4]).astype(np.float32)))
    # oracle spot check on rows of the full-size frame would take minutes; the 64x64 golden ids cover the hit path


# ---- errors -----------------------------------------------------------------------------------------------------------
def test_error_reporting(backend, cornell):
    bad = SceneData(cornell.vertices, cornell.indices, cornell.submeshes.copy(), cornell.index_counts, cornell.xforms, cornell.materials)
    bad.submeshes[0, 3] = 99
    with pytest.raises(BackendError) as e:
        backend.scene_upload(bad)
    assert e.value.code == 1 and "material" in str(e.value)
    bad2 = SceneData(cornell.vertices[:100], cornell.indices, cornell.submeshes, cornell.index_counts, cornell.xforms, cornell.materials)
    with pytest.raises(BackendError):
        backend.scene_upload(bad2)
    backend.scene_upload(cornell)
    with pytest.raises(BackendError):
        backend.trace(1)                 # hierarchy not built after a new upload
    backend.build_accel()
    with pytest.raises(BackendError):
        backend.trace(0)                 # frame index is 1-based
    with pytest.raises(BackendError):
        backend.set_params(Params.reference(spp_per_frame=0))
    with pytest.raises(BackendError):
        backend.resize(0, 4)
    # entry distances are ordered by their bit patterns: a negative (or NaN) tmin is refused, not mis-sorted
    for tmin in (-1e-3, float("nan")):
        with pytest.raises(BackendError) as e:
            backend.set_params(Params.reference(tmin=tmin))
        assert "tmin" in str(e.value)
        rays = random_rays(np.random.default_rng(0), 4, -0.2, 0.2, tmin=tmin)
        with pytest.raises(BackendError):
            backend.debug_intersect(rays)
    # a table with a non-zero count must not be NULL
    import ctypes as C
    desc, keep = cornell.as_desc()
    desc.num_submeshes = 0
    desc.vertices = C.POINTER(C.c_float)()
    assert backend.lib.pt_scene_upload(backend.h, C.byref(desc)) == 1
    desc, keep = cornell.as_desc()
    desc.num_textures = 2
    desc.textures = C.POINTER(type(keep[0]))()
    assert backend.lib.pt_scene_upload(backend.h, C.byref(desc)) == 1
    backend.scene_upload(cornell)
    backend.build_accel()


# ---- compute passes around the path tracer ---------------------------------------------------------------------
@pytest.mark.parametrize("fn", [5, 6, 7, 8])
def test_det_math_of_the_passes_bit_exact(backend, fn):
    rng = np.random.default_rng(40 + fn)
    n = 300000
    x = (rng.normal(size=n) * np.exp(rng.uniform(-25, 15, n))).astype(np.float32)
    y = (rng.normal(size=n) * np.exp(rng.uniform(-25, 15, n))).astype(np.float32)
    if fn == 6:
        x = rng.uniform(-1.001, 1.001, n).astype(np.float32)
    if fn == 7:
        x = rng.uniform(-100, 100, n).astype(np.float32)
    x[:10] = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 65520.0, 2.0 ** -25, 1e-45]
    y[:10] = [0.0, -0.0, 0.0, -0.0, np.inf, 1.0, 1.0, np.nan, -np.inf, -1e-45]
    got = backend.debug_math(fn, x, y if fn == 5 else None)
    want = oracle_lib.det_math(fn, x, y)
    assert np.array_equal(bits(got), bits(want))


def test_postprocess_bit_exact(backend, cornell, cameras):
    o, rays = render_pair(backend, cornell, cameras["512x512"], 200, 120, (1, 2, 3), spp_per_frame=2, max_bounces=4)
    for exposure in (0.8, 0.0, 7.5):
        got = backend.postprocess(exposure)
        want = oracle_lib.postprocess(o.read_image(), exposure)
        assert got.shape == (120, 200, 4) and np.array_equal(got, want)
    # arbitrary radiance, including what a renderer can produce by accident
    rng = np.random.default_rng(3)
    img = np.exp(rng.uniform(-12, 8, (120, 200, 4))).astype(np.float32)
    img[0, :6, 0] = [np.nan, np.inf, -np.inf, -1.0, 0.0, 1e30]
    backend.write_accum(np.concatenate([img[..., :3], np.ones((120, 200, 1), np.float32)], -1))
    backend.resolve()
    assert np.array_equal(backend.postprocess(1.25), oracle_lib.postprocess(backend.read_image(), 1.25))


@pytest.mark.parametrize("w,h,n", [(64, 32, 16), (37, 19, 33), (1, 1, 1), (512, 256, 128)])
def test_equirect_to_cube_bit_exact(backend, cornell, w, h, n):
    backend.scene_upload(cornell)
    rng = np.random.default_rng(w + n)
    img = np.exp(rng.uniform(-6, 6, (h, w, 4))).astype(np.float32)
    backend.env_from_equirect(img, n)
    got = backend.read_env()
    want = oracle_lib.equirect_to_cube(img, n)
    assert got.shape == (6, n, n, 4) and np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("params", [(3.14, 0.0, 0.0), (2.0, 1.1, 0.7), (8.0, 5.0, 1.5)])
def test_preetham_sky_bit_exact(backend, cornell, params):
    backend.scene_upload(cornell)
    backend.env_preetham(*params, face_size=96)
    assert np.array_equal(bits(backend.read_env()), bits(oracle_lib.preetham_sky(*params, 96)))


def test_tracing_under_a_converted_environment(backend, cornell, cameras):
    """Environment made on the GPU from a lat-long image, then path traced: same bits as the oracle tracing
    under the oracle's conversion of the same image."""
    import copy
    rng = np.random.default_rng(9)
    img = np.exp(rng.uniform(-2, 3, (32, 64, 4))).astype(np.float32)
    sc = copy.copy(cornell)
    sc.env = oracle_lib.equirect_to_cube(img, 32)
    W, H = 96, 80
    p = Params.reference(spp_per_frame=2, max_bounces=4)
    backend.scene_upload(cornell)
    backend.build_accel()
    backend.env_from_equirect(img, 32)
    backend.resize(W, H)
    backend.set_camera(cameras["512x512"])
    backend.set_params(p)
    o = oracle_lib.Oracle(sc)
    o.resize(W, H)
    o.set_camera(cameras["512x512"])
    o.set_params(p)
    for f in (1, 2, 3):
        backend.trace(f)
        o.trace(f)
    assert np.array_equal(bits(backend.read_image()), bits(o.read_image()))
    with pytest.raises(BackendError):
        backend.env_preetham(3.0, 0.0, 0.0, face_size=0)


def test_transform_edit_rebuilds_the_hierarchy(backend, cornell, cameras):
    """RayTracingLayer.cpp:569-604: a submesh is moved / scaled and the acceleration structure recreated.  Same bits
    as the oracle on the moved scene, and as a fresh upload of it."""
    import copy
    W, H = 160, 120
    moved = copy.copy(cornell)
    x = np.array(cornell.xforms, np.float32).reshape(-1, 12).copy()
    k = len(x) - 1
    m = x[k].reshape(3, 4)
    m[:, 3] += np.array([0.03, 0.05, -0.02], np.float32)          # translate the last submesh
    m[:, :3] *= np.float32(1.25)                                     # and scale it
    x[0].reshape(3, 4)[0, 3] -= np.float32(0.01)
    moved.xforms = x
    o, rays = render_pair(backend, cornell, cameras["512x512"], W, H, (1, 2), spp_per_frame=1, max_bounces=4)
    before = backend.read_image()
    st = backend.update_transforms(x)
    assert st.num_triangles == 63978 and st.num_nodes > 0
    o2, rays2 = render_pair(backend, moved, cameras["512x512"], W, H, (1, 2), upload=False, spp_per_frame=1, max_bounces=4)
    assert_same_frame(backend, o2, rays2)
    assert not np.array_equal(backend.read_image(), before)
    with pytest.raises(BackendError):
        backend.update_transforms(x[:-1])
    bad = x.copy()
    bad[0, 0] = np.nan
    with pytest.raises(BackendError):
        backend.update_transforms(bad)


UPDATE_REFIT = 0x100


def test_transform_edit_with_refit(backend, cornell, cameras):
    """PT_UPDATE_REFIT keeps the hierarchy's topology and recomputes triangles, boxes and node grids: refitting to
    the transforms the hierarchy was built with gives back the same bytes; after a move the frames equal the
    oracle's on the moved scene (the closest hit does not depend on the hierarchy), several edits in a row too;
    and a rebuild afterwards is as good as new."""
    import copy
    W, H = 160, 120
    backend.scene_upload(cornell)
    st0 = backend.build_accel(2)      # no splitting: a refit bounds every reference of a split triangle by the whole triangle
    nodes0, tris0 = backend.debug_read_bvh()
    st = backend.update_transforms(cornell.xforms, UPDATE_REFIT)
    nodes1, tris1 = backend.debug_read_bvh()
    assert np.array_equal(nodes0, nodes1) and np.array_equal(tris0, tris1)
    assert (st.num_nodes, st.num_leaves, st.max_depth) == (st0.num_nodes, st0.num_leaves, st0.max_depth)
    assert np.isclose(st.sah_cost, st0.sah_cost, rtol=1e-5)
    x = np.array(cornell.xforms, np.float32).reshape(-1, 12).copy()
    for step in range(3):
        k = len(x) - 1 - step
        m = x[k].reshape(3, 4)
        m[:, 3] += np.array([0.03, 0.05 * (step + 1), -0.02], np.float32)
        m[:, :3] *= np.float32(1.25 - 0.3 * step)
        st = backend.update_transforms(x, UPDATE_REFIT)
        assert st.num_nodes == st0.num_nodes
        moved = copy.copy(cornell)
        moved.xforms = x.copy()
        o, rays = render_pair(backend, moved, cameras["512x512"], W, H, (1, 2), upload=False, spp_per_frame=1, max_bounces=4)
        assert_same_frame(backend, o, rays)
        rng = np.random.default_rng(40 + step)
        r = random_rays(rng, 20000, [-0.35, -0.05, -0.6], [0.3, 0.6, 0.8])
        hg, _ = backend.debug_intersect(r)
        assert np.array_equal(hg, o.intersect(r, use_bvh=True))
    refit_cost = st.sah_cost
    st = backend.update_transforms(x)              # rebuild over the same transforms
    assert st.sah_cost <= refit_cost * 1.0001
    fresh = copy.copy(cornell)
    fresh.xforms = x
    backend.scene_upload(fresh)
    st2 = backend.build_accel()
    assert np.isclose(st.sah_cost, st2.sah_cost, rtol=1e-5)
    backend.scene_upload(SceneData(cornell.vertices[:3], np.arange(3, dtype=np.uint32), [(0, 0, 0, 0)], [3],
                                   cornell.xforms[:1], cornell.materials))
    with pytest.raises(BackendError):
        backend.update_transforms(cornell.xforms[:1], UPDATE_REFIT)        # nothing built yet for this scene


def _soup(tris):
    """SceneData of independent triangles (n, 3, 3), one submesh, identity transform."""
    n = len(tris)
    v = np.zeros((3 * n, 12), np.float32)
    v[:, 0:3] = np.asarray(tris, np.float32).reshape(-1, 3)
    v[:, 5:8] = (0, 0, 1)
    ident = np.array([[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]], np.float32)
    return SceneData(v, np.arange(3 * n, dtype=np.uint32), [(0, 0, 0, 0)], [3 * n], ident, [make_material()])


@pytest.mark.parametrize("kind", ["duplicates", "line", "scales", "slivers", "zero_area", "huge", "nonfinite"])
def test_degenerate_distributions_build_and_trace(backend, kind):
    """Inputs that make spatial orders degenerate: the build must neither fail nor produce a hierarchy that is too
    deep for the traversal stack, and traversal must still equal the brute-force loop."""
    rng = np.random.default_rng(11)
    n = 60000
    base = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    if kind == "duplicates":                       # the same triangle many times, plus a few others
        tris = np.repeat(base[None], n, 0)
        tris[: n // 100] += rng.uniform(-3, 3, (n // 100, 1, 3)).astype(np.float32)
    elif kind == "line":                           # a one-dimensional distribution
        tris = base[None] * 1e-3 + np.stack([np.arange(n) * 2e-3, np.zeros(n), np.zeros(n)], -1)[:, None, :].astype(np.float32)
    elif kind == "scales":                         # sizes over six orders of magnitude
        s = np.exp(rng.uniform(np.log(1e-3), np.log(1e3), n)).astype(np.float32)
        tris = base[None] * s[:, None, None] + rng.uniform(-50, 50, (n, 1, 3)).astype(np.float32)
    elif kind == "zero_area":                      # points and segments among ordinary triangles
        tris = base[None] * 0.1 + rng.uniform(-5, 5, (n, 1, 3)).astype(np.float32)
        tris[::3, 1] = tris[::3, 0]
        tris[::7, 2] = tris[::7, 0]
    elif kind == "huge":                           # coordinates far beyond what squares without overflow
        tris = (base[None] + rng.uniform(-5, 5, (n, 1, 3))).astype(np.float32)
        tris[:50] *= np.float32(1e18)
        tris[50:100] *= np.float32(3e37)
    elif kind == "nonfinite":                      # a renderer gets handed these; they must hit nothing and break nothing
        tris = (base[None] + rng.uniform(-5, 5, (n, 1, 3))).astype(np.float32)
        tris[:200, 0, 0] = np.nan
        tris[200:400, 1, 2] = np.inf
        tris[400:600, 2, 1] = -np.inf
    else:                                          # long thin triangles crossing the whole scene
        a = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32) * 20
        tris = np.stack([a, a + d, a + d + rng.normal(size=(n, 3)).astype(np.float32) * 1e-2], 1)
    sc = _soup(tris.astype(np.float32))
    backend.scene_upload(sc)
    if kind == "huge":                             # refused with a message rather than traced wrongly
        with pytest.raises(BackendError, match="1e18"):
            backend.build_accel()
        return
    st = backend.build_accel()
    assert st.num_triangles == n and 7 * st.max_depth + 1 <= 1024
    finite = tris.reshape(-1, 3)[np.isfinite(tris.reshape(-1, 3)).all(1) & (np.abs(tris.reshape(-1, 3)) < 1e6).all(1)]
    lo, hi = finite.min(0), finite.max(0)
    rays = random_rays(rng, 3000, lo - 1, hi + 1)
    got, _ = backend.debug_intersect(rays)
    o = oracle_lib.Oracle(sc)
    want = o.intersect(rays, use_bvh=False)
    assert np.array_equal(got, want)
    assert (got[:, 3] != M32).sum() > 10 or kind == "line"


def test_scenes_of_one_to_twenty_triangles(backend):
    """Sizes around the treelet size (8 subtrees) and the leaf size (4): every builder variant equals the brute-force loop."""
    rng = np.random.default_rng(5)
    for n in range(1, 21):
        tris = (rng.uniform(-1, 1, (n, 3, 3)) + rng.uniform(-4, 4, (n, 1, 3))).astype(np.float32)
        sc = _soup(tris)
        backend.scene_upload(sc)
        rays = random_rays(rng, 400, [-5, -5, -5], [5, 5, 5])
        want = oracle_lib.Oracle(sc).intersect(rays, use_bvh=False)
        for flags in (0, 8, 16, 2):
            st = backend.build_accel(flags)
            assert st.num_triangles == n
            got, _ = backend.debug_intersect(rays)
            assert np.array_equal(got, want), (n, flags)
