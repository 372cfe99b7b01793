"""How much hierarchy quality is left: a top-down binned-SAH binary tree over a workload's triangles, built here in numpy,
priced with the same surface-area model the GPU builder's collapse minimises (c(root, 1) of k_collapse_dp: a wide node
costs its box area, a leaf of P <= 4 triangles area x (0.4 + 0.25 P), areas relative to the root's).  Compare the printed
cost with `config.hierarchy.sah_cost` of `bench.py --build-flags 2` (every triangle entered once) on the same workload.

usage: python scripts/sah_headroom.py [workload] [detail]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import procedural  # noqa: E402

C_NODE, C_TRI, C_LEAF, LEAF_MAX = 1.0, 0.25, 0.4, 4
BINS = 32


def world_triangles(sc):
    out = []
    for s in range(len(sc.submeshes)):
        vo, io, cnt = int(sc.submeshes[s, 1]), int(sc.submeshes[s, 2]), int(sc.index_counts[s])
        p = sc.vertices[sc.indices[io:io + cnt].astype(np.int64) + vo, :3].astype(np.float64)
        m = sc.xforms[s].reshape(3, 4).astype(np.float64)
        out.append((p @ m[:, :3].T + m[:, 3]).reshape(-1, 3, 3))
    return np.concatenate(out)


def area(lo, hi):
    e = np.maximum(hi - lo, 0.0)
    return e[..., 0] * e[..., 1] + e[..., 1] * e[..., 2] + e[..., 2] * e[..., 0]


def build(tlo, thi):
    """returns left, right (leaf t -> -(t + 1)), lo, hi, count of the internal nodes; children have larger numbers than parents"""
    cen = 0.5 * (tlo + thi)
    left, right, nlo, nhi, ncnt = [], [], [], [], []
    stack = [(np.arange(len(tlo)), -1, 0)]
    while stack:
        idx, parent, side = stack.pop()
        if len(idx) == 1:
            ref = -(int(idx[0]) + 1)
        else:
            ref = len(left)
            left.append(0); right.append(0)
            lo, hi = tlo[idx].min(0), thi[idx].max(0)
            nlo.append(lo); nhi.append(hi); ncnt.append(len(idx))
            c = cen[idx]
            clo, chi = c.min(0), c.max(0)
            best = (np.inf, -1, None)
            for ax in range(3):
                ext = chi[ax] - clo[ax]
                if not ext > 0:
                    continue
                b = np.minimum(((c[:, ax] - clo[ax]) * (BINS / ext)).astype(np.int64), BINS - 1)
                blo = np.full((BINS, 3), np.inf); bhi = np.full((BINS, 3), -np.inf)
                np.minimum.at(blo, b, tlo[idx]); np.maximum.at(bhi, b, thi[idx])
                cnt = np.bincount(b, minlength=BINS)
                llo = np.minimum.accumulate(blo, 0); lhi = np.maximum.accumulate(bhi, 0)
                rlo = np.minimum.accumulate(blo[::-1], 0)[::-1]; rhi = np.maximum.accumulate(bhi[::-1], 0)[::-1]
                lc = np.cumsum(cnt)
                cost = area(llo[:-1], lhi[:-1]) * lc[:-1] + area(rlo[1:], rhi[1:]) * (len(idx) - lc[:-1])
                cost[(lc[:-1] == 0) | (lc[:-1] == len(idx))] = np.inf
                k = int(np.argmin(cost))
                if cost[k] < best[0]:
                    best = (cost[k], ax, b <= k)
            if best[2] is None:                      # all centres equal: halve
                mask = np.arange(len(idx)) < len(idx) // 2
            else:
                mask = best[2]
            stack.append((idx[~mask], ref, 1))
            stack.append((idx[mask], ref, 0))
        if parent >= 0:
            (left if side == 0 else right)[parent] = ref
    return np.array(left), np.array(right), np.array(nlo), np.array(nhi), np.array(ncnt)


def collapse_cost(left, right, nlo, nhi, ncnt, tlo, thi):
    root_area = area(nlo[0], nhi[0])
    na = area(nlo, nhi) / root_area
    ta = area(tlo, thi) / root_area
    n = len(left)
    c = np.zeros((n, 8))                              # c[node, 1..7]

    def child(ref):
        if ref < 0:
            return np.full(8, ta[-ref - 1] * (C_LEAF + C_TRI))
        return c[ref]

    for i in range(n - 1, -1, -1):
        cl, cr = child(left[i]), child(right[i])
        split = np.full(9, np.inf)
        for j in range(2, 9):
            ks = np.arange(max(1, j - 7), min(7, j - 1) + 1)
            split[j] = (cl[ks] + cr[j - ks]).min()
        leaf = na[i] * (C_LEAF + ncnt[i] * C_TRI) if ncnt[i] <= LEAF_MAX else np.inf
        c[i, 1] = min(leaf, na[i] * C_NODE + split[8])
        for k in range(2, 8):
            c[i, k] = min(split[k], c[i, k - 1])
    return c[0, 1]


if __name__ == "__main__":
    workload = sys.argv[1] if len(sys.argv) > 1 else "sponza_class"
    detail = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    sc = procedural.sponza_class(detail=detail) if workload == "sponza_class" else \
        procedural.sponza_class(detail=detail, layout="sponza") if workload == "sponza_like" else procedural.cornell_like()
    tris = world_triangles(sc)
    tlo, thi = tris.min(1), tris.max(1)
    t0 = time.time()
    tree = build(tlo, thi)
    t1 = time.time()
    cost = collapse_cost(*tree, tlo, thi)
    print("%s detail %g: %d triangles, binned-SAH tree in %.0f s, collapse cost %.2f (in %.0f s)" % (workload, detail, len(tris), t1 - t0, cost, time.time() - t1))
