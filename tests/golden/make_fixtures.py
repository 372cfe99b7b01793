"""Mints the committed fixtures under tests/golden/.  Run in the authoring container, where the
reference checkout is mounted at /root/reference; the GPU box only reads the results.

  cornell_scene.npz   scene tables of PathTracer/assets/models/CornellBox.gltf (its embedded buffer)
                      with the reference's layer transform scale(0.1) (RayTracingLayer.cpp:32-34),
                      produced by the independent Python glTF reader tests/gltf_py.py.  Data only.
  cornell_cameras.npz CameraBuffer matrices (5 x mat4, column-major) of pathtracer_amd.camera.cornell_camera
                      for the sizes the tests use, so goldens do not depend on the host's LAPACK rounding.
  pcg_kat.json        first 16 outputs of PCG_Hash (Globals.h:54-59) for four seeds, computed with Python
                      integers (independent of the oracle).
  suzanne_scene.npz   scene tables of PathTracer/assets/models/Suzanne/glTF/Suzanne.gltf (3 936 triangles with tangents
                      and uv) and its two PNG textures decoded with Pillow (RGBA8, 1024 x 1024 each; base colour and
                      metallic-roughness), by the independent Python reader.  Data only.
  suzanne_cameras.npz CameraBuffer matrices of pathtracer_amd.camera.suzanne_camera for the sizes the tests use.
  cornell_primary_ids_64.npy      (submesh, primitive) of the primary hit per pixel, 64x64, from the
                      oracle's BRUTE-FORCE intersector (no hierarchy involved).
  cornell_radiance_128_16spp.npy  accumulation image after frames 1..17 (16 accumulated samples),
                      128x128, 1 spp/frame, 4 bounces, from the oracle.
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


def suzanne():
    from PIL import Image
    base = os.path.join(REF, "Suzanne", "glTF")
    sc = gltf_py.load_gltf(os.path.join(base, "Suzanne.gltf"), use_textures=True)
    import json
    g = json.load(open(os.path.join(base, "Suzanne.gltf")))
    sc.textures = [np.ascontiguousarray(np.asarray(Image.open(os.path.join(base, im["uri"])).convert("RGBA"), np.uint8)) for im in g["images"]]
    sc.texture_flags = [0] * len(sc.textures)          # the asset's one sampler object is empty: bilinear, repeat
    sc.save(os.path.join(HERE, "suzanne_scene.npz"))
    print("suzanne_scene.npz:", sc.num_triangles, "triangles,", [t.shape for t in sc.textures])


def cameras():
    from pathtracer_amd import camera as cam_
    np.savez(os.path.join(HERE, "suzanne_cameras.npz"),
             **{"%dx%d" % wh: cam_.suzanne_camera(*wh) for wh in [(1920, 1080), (240, 135), (96, 54)]})
    from pathtracer_amd import camera
    np.savez(os.path.join(HERE, "cornell_cameras.npz"),
             **{"%dx%d" % wh: camera.cornell_camera(*wh) for wh in [(64, 64), (128, 128), (512, 512), (1920, 1080), (96, 64), (24, 16), (16, 16)]})


def pcg():
    import json
    M = 0xFFFFFFFF
    out = {}
    for s0 in (0, 1, 12345, 0xFFFFFFFF):
        seq, s = [], s0
        for _ in range(16):
            s = (s * 747796405 + 2891336453) & M
            w = ((((s >> ((s >> 28) + 4)) ^ s) & M) * 277803737) & M
            seq.append(((w >> 22) ^ w) & M)
        out[str(s0)] = seq
    json.dump(out, open(os.path.join(HERE, "pcg_kat.json"), "w"), indent=0)


def oracle_goldens():
    from pathtracer_amd.scene import Params, SceneData
    from tests import oracle_lib
    sc = SceneData.load(os.path.join(HERE, "cornell_scene.npz"))
    cams = np.load(os.path.join(HERE, "cornell_cameras.npz"))
    o = oracle_lib.Oracle(sc)
    o.resize(64, 64)
    o.set_camera(cams["64x64"])
    o.set_params(Params.reference(spp_per_frame=1, max_bounces=1))
    o.trace(1, use_bvh=False)
    np.save(os.path.join(HERE, "cornell_primary_ids_64.npy"), o.read_primary_ids())
    o.resize(128, 128)
    o.set_camera(cams["128x128"])
    o.set_params(Params.reference(spp_per_frame=1, max_bounces=4))
    for f in range(1, 18):
        o.trace(f)
    np.save(os.path.join(HERE, "cornell_radiance_128_16spp.npy"), o.read_accum())


if __name__ == "__main__":
    if os.path.exists(REF):
        cornell()
        suzanne()
    cameras()
    pcg()
    oracle_goldens()
