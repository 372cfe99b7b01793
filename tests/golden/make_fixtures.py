"""Mints the committed fixtures under tests/golden/.  Run in the authoring container, where the
reference checkout is mounted at /root/reference; the GPU box only reads the results.

  cornell_scene.npz   scene tables of PathTracer/assets/models/CornellBox.gltf (its embedded buffer)
                      with the reference's layer transform scale(0.1) (RayTracingLayer.cpp:32-34),
                      produced by the independent Python glTF reader tests/gltf_py.py.  Data only.
  (further fixtures are appended by the sections below)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests import gltf_py  # noqa: E402

REF = "/root/reference/PathTracer/assets/models"


def cornell():
    layer = np.diag([0.1, 0.1, 0.1, 1.0])
    sc = gltf_py.load_gltf(os.path.join(REF, "CornellBox.gltf"), layer)
    sc.save(os.path.join(HERE, "cornell_scene.npz"))
    print("cornell_scene.npz:", sc.num_triangles, "triangles")


if __name__ == "__main__":
    cornell()
