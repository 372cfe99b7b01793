"""Prints the shape of the built hierarchy: children per node, triangles per leaf, nodes per level."""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np
from pathtracer_amd import procedural
from pathtracer_amd.backend import Backend
from pathtracer_amd.scene import SceneData

name = sys.argv[1] if len(sys.argv) > 1 else "sponza_class"
if name == "cornell":
    sc = SceneData.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "cornell_scene.npz"))
else:
    sc = getattr(procedural, name)()
b = Backend(0)
b.scene_upload(sc)
st = b.build_accel()
nodes, tris = b.debug_read_bvh()
nodes = nodes.reshape(-1, 32)
child = nodes[:, 4:12]
empty = child == 0xFFFFFFFF
leaf = ((child >> 31) == 1) & ~empty
inner = ~empty & ~leaf
cnt = ((child >> 28) & 7) + 1
print(name, "triangles", st.num_triangles, "nodes", len(nodes), "leaves", int(leaf.sum()), "levels", st.max_depth, "sah", st.sah_cost)
print("children per node histogram (1..8):", np.bincount((~empty).sum(1), minlength=9)[1:].tolist(), "mean %.2f" % (~empty).sum(1).mean())
print("inner children per node mean %.2f, leaf children per node mean %.2f" % (inner.sum(1).mean(), leaf.sum(1).mean()))
print("triangles per leaf histogram (1..):", np.bincount(cnt[leaf])[1:].tolist(), "mean %.2f" % cnt[leaf].mean())
