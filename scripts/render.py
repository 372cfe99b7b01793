"""Renders a glTF file to a PPM through the host layer: import, textures, environment, progressive frames,
post-processing.  Example:
  python scripts/render.py /path/to/Suzanne.gltf out.ppm --frames 64 --eye 0 0.3 3.2 --target 0 0 0 --hdr sky.hdr
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pathtracer_amd import host
from pathtracer_amd.scene import Params


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("gltf")
    ap.add_argument("out")
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--spp", type=int, default=5)            # the reference's SAMPLE_COUNT
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--scale", type=float, default=1.0, help="layer transform (the reference scales the Cornell box by 0.1)")
    ap.add_argument("--eye", type=float, nargs=3, default=(0.0, 0.5, 3.0))
    ap.add_argument("--target", type=float, nargs=3, default=(0.0, 0.0, 0.0))
    ap.add_argument("--hdr", default=None, help="Radiance .hdr environment")
    ap.add_argument("--exposure", type=float, default=0.8)
    args = ap.parse_args()

    src = host.MeshSource(args.gltf)
    print("decoded %d texture(s)" % src.load_textures())
    layer = host.RayTracingLayer(src, transform=np.diag([args.scale] * 3 + [1.0]).astype(np.float32),
                                 camera=host.camera_spec(position=tuple(args.eye), target=tuple(args.target)))
    layer.SetParams(Params.reference(spp_per_frame=args.spp, max_bounces=args.bounces))
    layer.SetViewportSize(args.width, args.height)
    layer.SetPostProcessing(True, args.exposure)
    if args.hdr:
        layer.SetRadianceMap(args.hdr)
    for _ in range(args.frames):
        layer.OnUpdate()
        layer.OnRender()
    host.write_ppm(args.out, layer.ReadPostProcessed())
    print("%s: %dx%d, %d frames x %d spp, %.1f M rays" % (args.out, args.width, args.height, args.frames, args.spp, layer.RayCount() / 1e6))
    layer.close()


if __name__ == "__main__":
    main()
