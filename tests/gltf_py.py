"""Independent glTF 2.0 reader in Python (json + numpy), used by tests to check the C++ host loader
and to mint scene fixtures.  Same conventions as pathtracer_amd/host/gltf.cpp:
  * scene nodes are walked depth first in file order; every primitive of a node's mesh is one submesh
    (one TLAS instance in the reference, ClosestHit.glsl:147-150);
  * vertices are 12 floats (pos3 uv2 nrm3 tan4, ClosestHit.glsl:56-81), indices uint32, one shared
    vertex buffer and one index buffer with per-submesh offsets;
  * materials are the 22-word records of ClosestHit.glsl:113-142: glTF defaults (base colour 1,
    metallic 1, roughness 1, emissive 0, emissive strength 1), KHR_materials_ior else 1.5, Disney
    extras 0, texture indices = glTF image indices or -1.
"""
import base64
import json
import os

import numpy as np

from pathtracer_amd.scene import SceneData, make_material

_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _node_local(node):
    if "matrix" in node:
        return np.array(node["matrix"], np.float64).reshape(4, 4).T
    t = np.array(node.get("translation", [0, 0, 0]), np.float64)
    q = np.array(node.get("rotation", [0, 0, 0, 1]), np.float64)
    s = np.array(node.get("scale", [1, 1, 1]), np.float64)
    x, y, z, w = q
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    m = np.eye(4)
    m[:3, :3] = r * s[None, :]
    m[:3, 3] = t
    return m


def _sheen(e):
    """sheenColorFactor -> Disney sheen strength (brightest channel) and tint (saturation)."""
    if not e:
        return {}
    c = np.asarray(e.get("sheenColorFactor", [0, 0, 0]), np.float32)
    hi, lo = c.max(), c.min()
    return {"sheen": hi, "sheen_tint": (hi - lo) / hi if hi > 0 else np.float32(0)}


def load_gltf(path, layer_transform=None, use_textures=False):
    with open(path) as f:
        g = json.load(f)
    base = os.path.dirname(path)
    buffers = []
    for b in g["buffers"]:
        uri = b["uri"]
        if uri.startswith("data:"):
            buffers.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            with open(os.path.join(base, uri), "rb") as f:
                buffers.append(f.read())

    def accessor(i):
        a = g["accessors"][i]
        bv = g["bufferViews"][a["bufferView"]]
        dt = np.dtype(_COMP[a["componentType"]])
        nc = _NCOMP[a["type"]]
        off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = bv.get("byteStride", 0) or dt.itemsize * nc
        raw = np.frombuffer(buffers[bv["buffer"]], np.uint8, count=(a["count"] - 1) * stride + dt.itemsize * nc, offset=off)
        rows = np.lib.stride_tricks.as_strided(raw, (a["count"], dt.itemsize * nc), (stride, 1))
        return np.ascontiguousarray(rows).view(dt).reshape(a["count"], nc)

    layer = np.eye(4) if layer_transform is None else np.asarray(layer_transform, np.float64)
    mats = []
    for m in g.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        ext = m.get("extensions", {})

        def tex(ref):
            if not use_textures or ref is None:
                return -1
            return g["textures"][ref["index"]].get("source", -1)
        mats.append(make_material(
            albedo=pbr.get("baseColorFactor", [1, 1, 1, 1])[:3], metallic=pbr.get("metallicFactor", 1.0),
            roughness=pbr.get("roughnessFactor", 1.0), emissive=m.get("emissiveFactor", [0, 0, 0]),
            emissive_strength=ext.get("KHR_materials_emissive_strength", {}).get("emissiveStrength", 1.0),
            albedo_map=tex(pbr.get("baseColorTexture")), mr_map=tex(pbr.get("metallicRoughnessTexture")),
            normal_map=tex(m.get("normalTexture")), ior=ext.get("KHR_materials_ior", {}).get("ior", 1.5),
            spec_trans=ext.get("KHR_materials_transmission", {}).get("transmissionFactor", 0.0),
            clearcoat=ext.get("KHR_materials_clearcoat", {}).get("clearcoatFactor", 0.0),
            clearcoat_roughness=ext.get("KHR_materials_clearcoat", {}).get("clearcoatRoughnessFactor", 0.0),
            anisotropic=ext.get("KHR_materials_anisotropy", {}).get("anisotropyStrength", 0.0),
            **_sheen(ext.get("KHR_materials_sheen"))))
    default_mat = None
    verts, inds, subs, counts, xfs, names = [], [], [], [], [], []
    nv = ni = 0

    def walk(ni_, parent):
        nonlocal nv, ni, default_mat
        node = g["nodes"][ni_]
        world = parent @ _node_local(node)
        if "mesh" in node:
            mesh = g["meshes"][node["mesh"]]
            for pi, prim in enumerate(mesh["primitives"]):
                if prim.get("mode", 4) != 4:
                    continue
                at = prim["attributes"]
                pos = accessor(at["POSITION"]).astype(np.float32)
                n = len(pos)
                v = np.zeros((n, 12), np.float32)
                v[:, 0:3] = pos
                if "TEXCOORD_0" in at:
                    v[:, 3:5] = accessor(at["TEXCOORD_0"]).astype(np.float32)
                if "NORMAL" in at:
                    v[:, 5:8] = accessor(at["NORMAL"]).astype(np.float32)
                if "TANGENT" in at:
                    v[:, 8:12] = accessor(at["TANGENT"]).astype(np.float32)
                else:
                    v[:, 8:12] = (1, 0, 0, 1)
                idx = accessor(prim["indices"]).astype(np.uint32).reshape(-1) if "indices" in prim else np.arange(n, dtype=np.uint32)
                mi = prim.get("material")
                if mi is None:
                    if default_mat is None:
                        default_mat = len(mats)
                        mats.append(make_material())
                    mi = default_mat
                subs.append((0, nv, ni, mi))
                counts.append(len(idx) - len(idx) % 3)
                m = (layer @ world)[:3, :]
                xfs.append(m.astype(np.float32).reshape(12))
                names.append(node.get("name", mesh.get("name", "node%d" % ni_)) + ("" if pi == 0 else ".%d" % pi))
                verts.append(v)
                inds.append(idx)
                nv += n
                ni += len(idx)
        for c in node.get("children", []):
            walk(c, world)

    scene = g["scenes"][g.get("scene", 0)]
    for r in scene["nodes"]:
        walk(r, np.eye(4))
    return SceneData(np.concatenate(verts), np.concatenate(inds), np.array(subs, np.uint32), np.array(counts, np.uint32),
                     np.array(xfs, np.float32), np.array(mats, np.float32), names=names)
