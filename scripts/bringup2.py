import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd.backend import Backend
from tests import oracle_lib
b = Backend(0)
rng = np.random.default_rng(2)
n = 20000
def rand_surf(n, kind):
    s = np.zeros((n, 17), np.float32)
    s[:, 0:3] = rng.uniform(0.05, 1, (n, 3))
    s[:, 4] = rng.uniform(0.02, 1, n)
    s[:, 12] = 1.5
    if kind == 'diffuse':
        pass
    elif kind == 'metal':
        s[:, 3] = 1.0
    elif kind == 'mixed':
        s[:, 3] = rng.uniform(0, 1, n)
    elif kind == 'full':
        s[:, 3] = rng.uniform(0, 1, n); s[:, 5:12] = rng.uniform(0, 1, (n, 7)); s[:, 12] = rng.uniform(1.1, 2.0, n); s[:, 16] = rng.uniform(0, 1, n)
    aspect = np.sqrt(np.float32(1) - s[:, 16] * np.float32(0.9)).astype(np.float32)
    s[:, 13] = np.maximum(np.float32(0.001), s[:, 4] / aspect)
    s[:, 14] = np.maximum(np.float32(0.001), s[:, 4] * aspect)
    inside = rng.uniform(0, 1, n) < 0.3
    s[:, 15] = np.where(inside, np.float32(1) / s[:, 12], s[:, 12])
    return s
for kind in ['diffuse', 'metal', 'mixed', 'full']:
    s = rand_surf(n, kind)
    N = rng.normal(size=(n, 3)).astype(np.float32); N /= np.linalg.norm(N, axis=1, keepdims=True)
    V = rng.normal(size=(n, 3)).astype(np.float32); V /= np.linalg.norm(V, axis=1, keepdims=True)
    flip = (V * N).sum(1) < 0; N[flip] *= -1
    seeds = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    fg, Lg, pg, sg = b.debug_disney_sample(s, V, N, seeds)
    fc, Lc, pc, sc_ = oracle_lib.disney_sample(s, V, N, seeds)
    eqL = (Lg.view(np.uint32) == Lc.view(np.uint32)).all(1); eqf = (fg.view(np.uint32) == fc.view(np.uint32)).all(1); eqp = pg.view(np.uint32) == pc.view(np.uint32)
    print(kind, "sample: L eq %d/%d f eq %d pdf eq %d seeds eq %s" % (eqL.sum(), n, eqf.sum(), eqp.sum(), np.array_equal(sg, sc_)), flush=True)
    bad = np.where(~eqL)[0][:3]
    for i in bad: print("   L", Lg[i], Lc[i])
    L = rng.normal(size=(n, 3)).astype(np.float32); L /= np.linalg.norm(L, axis=1, keepdims=True)
    fg, pg = b.debug_disney_eval(s, V, N, L); fc, pc = oracle_lib.disney_eval(s, V, N, L)
    eqf = (fg.view(np.uint32) == fc.view(np.uint32)).all(1); eqp = pg.view(np.uint32) == pc.view(np.uint32)
    print(kind, "eval: f eq %d/%d pdf eq %d" % (eqf.sum(), n, eqp.sum()), flush=True)
    bad = np.where(~eqf)[0][:3]
    for i in bad: print("   f", fg[i], fc[i], pg[i], pc[i])
