"""The C++ host layer: glTF import against the independent Python reader, camera matrices, and
(on a GPU) the reference's frame sequencing through RayTracingLayer."""
import os
import re

import numpy as np
import pytest

from pathtracer_amd import camera, host
from pathtracer_amd.scene import Params
from tests import gltf_py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MODELS = "/root/reference/PathTracer/assets/models"
have_ref = os.path.exists(REF_MODELS)


def test_host_library_exports_every_declared_symbol():
    text = open(os.path.join(ROOT, "include", "pathtracer_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(pth_[a-z_0-9]+)\s*\(", text)))
    lib = host.load_library()
    for name in declared:
        assert hasattr(lib, name)
    assert declared == sorted(host.EXPORTS)


def write_gltf(tmp_path):
    """A two-node glTF with a child transform, uint16 indices, an external .bin and a default material."""
    import json
    import struct
    pos = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0)]
    nrm = [(0, 0, 1)] * 4
    idx = [0, 1, 2, 2, 1, 3]
    blob = b"".join(struct.pack("<3f", *p) for p in pos) + b"".join(struct.pack("<3f", *n) for n in nrm) + struct.pack("<6H", *idx)
    (tmp_path / "quad.bin").write_bytes(blob)
    g = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
        "nodes": [{"name": "root", "translation": [1, 2, 3], "scale": [2, 2, 2], "children": [1], "mesh": 0},
                  {"name": "kid", "rotation": [0, 0.7071068, 0, 0.7071068], "mesh": 0}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1}, "indices": 2, "material": 0},
                                   {"attributes": {"POSITION": 0, "NORMAL": 1}, "indices": 2}]}],
        "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.4, 0.6, 1], "metallicFactor": 0.25, "roughnessFactor": 0.5},
                       "emissiveFactor": [1, 0.5, 0.25],
                       "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 3.0}, "KHR_materials_ior": {"ior": 1.33},
                                      "KHR_materials_transmission": {"transmissionFactor": 0.75},
                                      "KHR_materials_clearcoat": {"clearcoatFactor": 0.5, "clearcoatRoughnessFactor": 0.125},
                                      "KHR_materials_sheen": {"sheenColorFactor": [0.5, 0.25, 0.125]},
                                      "KHR_materials_anisotropy": {"anisotropyStrength": 0.375}}}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"},
                      {"bufferView": 1, "componentType": 5126, "count": 4, "type": "VEC3"},
                      {"bufferView": 2, "componentType": 5123, "count": 6, "type": "SCALAR"}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 48}, {"buffer": 0, "byteOffset": 48, "byteLength": 48},
                        {"buffer": 0, "byteOffset": 96, "byteLength": 12}],
        "buffers": [{"uri": "quad.bin", "byteLength": 108}],
    }
    path = tmp_path / "quad.gltf"
    path.write_text(json.dumps(g))
    return str(path)


def hostile_gltfs(tmp_path):
    """Structurally valid JSON whose numbers aim at the loader's arithmetic: each must be refused, not read out of bounds."""
    import base64
    import copy
    import json
    import struct
    pos = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0)]
    blob = (b"".join(struct.pack("<3f", *p) for p in pos) + b"".join(struct.pack("<3f", 0, 0, 1) for _ in pos) +
            b"".join(struct.pack("<2f", p[0], p[1]) for p in pos) + b"".join(struct.pack("<4f", 1, 0, 0, 1) for _ in pos) + struct.pack("<6H", 0, 1, 2, 2, 1, 3))
    good = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0, "children": [1]}, {"mesh": 0}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2, "TANGENT": 3}, "indices": 4}]}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"}, {"bufferView": 1, "componentType": 5126, "count": 4, "type": "VEC3"},
                      {"bufferView": 2, "componentType": 5126, "count": 4, "type": "VEC2"}, {"bufferView": 3, "componentType": 5126, "count": 4, "type": "VEC4"},
                      {"bufferView": 4, "componentType": 5123, "count": 6, "type": "SCALAR"}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 48}, {"buffer": 0, "byteOffset": 48, "byteLength": 48},
                        {"buffer": 0, "byteOffset": 96, "byteLength": 32}, {"buffer": 0, "byteOffset": 128, "byteLength": 64},
                        {"buffer": 0, "byteOffset": 192, "byteLength": 12}],
        "buffers": [{"uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode(), "byteLength": len(blob)}],
    }

    def variant(edit):
        g = copy.deepcopy(good)
        edit(g)
        return g
    bad = {
        "scalar_normal": variant(lambda g: g["accessors"][1].update(type="SCALAR", bufferView=4, componentType=5123, count=6)),   # 3 components read from a 2-byte element
        "scalar_tangent_at_end": variant(lambda g: g["accessors"][3].update(type="SCALAR", bufferView=4, componentType=5123, count=6)),
        "vec2_position": variant(lambda g: g["accessors"][0].update(type="VEC2")),
        "negative_view_offset": variant(lambda g: g["bufferViews"][0].update(byteOffset=-16)),
        "negative_accessor_offset": variant(lambda g: g["accessors"][0].update(byteOffset=-8)),
        "fractional_offset": variant(lambda g: g["accessors"][0].update(byteOffset=0.5)),
        "huge_count": variant(lambda g: g["accessors"][0].update(count=2 ** 61)),
        "count_times_stride_wraps": variant(lambda g: (g["accessors"][0].update(count=2 ** 60 + 1), g["bufferViews"][0].update(byteStride=16))),
        "stride_below_element": variant(lambda g: g["bufferViews"][0].update(byteStride=4)),
        "huge_offset": variant(lambda g: g["bufferViews"][2].update(byteOffset=1e300)),
        "self_child": variant(lambda g: g["nodes"][1].update(children=[1])),
        "cycle": variant(lambda g: g["nodes"][1].update(children=[0])),
        "negative_accessor": variant(lambda g: g["meshes"][0]["primitives"][0]["attributes"].update(POSITION=-1)),
        "negative_view": variant(lambda g: g["accessors"][0].update(bufferView=-3)),
        "bad_buffer": variant(lambda g: g["bufferViews"][0].update(buffer=7)),
        "vec3_indices": variant(lambda g: g["accessors"][4].update(type="VEC3", count=2)),
        "bad_scene": variant(lambda g: g.update(scene=-1)),
        "bad_mesh": variant(lambda g: g["nodes"][0].update(mesh=1e9)),
    }
    deep = copy.deepcopy(good)
    deep["nodes"] = [{"children": [i + 1]} for i in range(400)] + [{"mesh": 0}]
    bad["too_deep"] = deep
    out = {}
    for name, g in [("good", good)] + list(bad.items()):
        path = tmp_path / ("hostile_%s.gltf" % name)
        path.write_text(json.dumps(g))
        out[name] = str(path)
    return out


def test_gltf_loader_refuses_hostile_accessors(tmp_path):
    files = hostile_gltfs(tmp_path)
    ok = host.MeshSource(files.pop("good")).scene_data()
    assert ok.num_triangles == 4 and np.array_equal(ok.vertices[:4, 8:12], np.tile([1, 0, 0, 1], (4, 1)))
    for name, path in files.items():
        with pytest.raises(RuntimeError):
            host.MeshSource(path)


def assert_same_scene(a, b):
    assert np.array_equal(a.vertices, b.vertices)
    assert np.array_equal(a.indices, b.indices)
    assert np.array_equal(a.submeshes, b.submeshes)
    assert np.array_equal(a.index_counts, b.index_counts)
    assert np.allclose(a.xforms, b.xforms, rtol=1e-6, atol=1e-7)
    assert np.array_equal(a.materials.view(np.uint32), b.materials.view(np.uint32))
    assert a.names == b.names


def test_gltf_import_synthetic(tmp_path):
    path = write_gltf(tmp_path)
    layer = np.diag([0.5, 0.5, 0.5, 1.0])
    got = host.MeshSource(path).scene_data(layer)
    want = gltf_py.load_gltf(path, layer)
    assert_same_scene(got, want)
    assert len(got.submeshes) == 4 and got.names == ["root", "root.1", "kid", "kid.1"]
    assert got.materials.shape[0] == 2                      # one declared + the default for the material-less primitive
    assert np.allclose(got.materials[0, :9], [0.2, 0.4, 0.6, 0.25, 0.5, 1, 0.5, 0.25, 3.0])
    assert np.isclose(got.materials[0, 21], 1.33) and np.isclose(got.materials[1, 21], 1.5)
    # child: parent TRS * 90 degree yaw, then the layer scale
    assert np.allclose(got.xforms[0].reshape(3, 4), [[1, 0, 0, 0.5], [0, 1, 0, 1], [0, 0, 1, 1.5]], atol=1e-6)
    assert np.allclose(got.xforms[2].reshape(3, 4), [[0, 0, 1, 0.5], [0, 1, 0, 1], [-1, 0, 0, 1.5]], atol=1e-6)


def test_gltf_sampler_state(tmp_path):
    """wrapS / wrapT / magFilter of a glTF sampler end up as the PT_TEX_* flags of the image the texture points at."""
    import json
    path = write_gltf(tmp_path)
    g = json.loads(open(path).read())
    g["images"] = [{"uri": "a.png"}, {"uri": "b.png"}, {"uri": "c.png"}]
    g["samplers"] = [{"wrapS": 33071, "wrapT": 33648, "magFilter": 9728}, {"wrapS": 33648}]
    g["textures"] = [{"source": 1, "sampler": 0}, {"source": 0, "sampler": 1}, {"source": 2}, {"source": 1, "sampler": 1}]
    g["materials"][0]["pbrMetallicRoughness"]["baseColorTexture"] = {"index": 0}
    open(path, "w").write(json.dumps(g))
    m = host.MeshSource(path)
    assert m.texture_uris() == ["a.png", "b.png", "c.png"]
    assert m.texture_samplers() == [2, 1 | (2 << 2) | 16, 0]      # image 1: the first texture that uses it decides


def test_gltf_import_errors(tmp_path):
    with pytest.raises(RuntimeError):
        host.MeshSource(str(tmp_path / "missing.gltf"))
    bad = tmp_path / "bad.gltf"
    bad.write_text("{ not json")
    with pytest.raises(RuntimeError):
        host.MeshSource(str(bad))


@pytest.mark.skipif(not have_ref, reason="reference assets are only mounted in the authoring container")
@pytest.mark.parametrize("rel", ["CornellBox.gltf", "Cube.gltf", "Suzanne/glTF/Suzanne.gltf"])
def test_gltf_import_reference_assets(rel):
    path = os.path.join(REF_MODELS, rel)
    layer = np.diag([0.1, 0.1, 0.1, 1.0])
    src = host.MeshSource(path)
    assert_same_scene(src.scene_data(layer), gltf_py.load_gltf(path, layer))


@pytest.mark.skipif(not have_ref, reason="reference assets are only mounted in the authoring container")
def test_suzanne_with_textures_equals_the_committed_fixture():
    """C++ import + C++ PNG decoder against the fixture minted by the Python reader + Pillow: tables, texels, sampler flags."""
    from pathtracer_amd.scene import SceneData
    src = host.MeshSource(os.path.join(REF_MODELS, "Suzanne/glTF/Suzanne.gltf"))
    with pytest.raises(RuntimeError):
        src.scene_data(with_textures=True)            # not decoded yet
    assert src.load_textures() == 2
    got = src.scene_data(with_textures=True)
    want = SceneData.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "suzanne_scene.npz"))
    assert_same_scene(got, want)
    assert got.texture_flags == want.texture_flags and len(got.textures) == 2
    assert all(np.array_equal(a, b) for a, b in zip(got.textures, want.textures))
    assert (src.scene_data().materials.view(np.int32)[:, 10:13] == -1).all()


@pytest.mark.skipif(not have_ref, reason="reference assets are only mounted in the authoring container")
def test_cornell_fixture_is_the_host_loader_output(cornell):
    got = host.MeshSource(os.path.join(REF_MODELS, "CornellBox.gltf")).scene_data(np.diag([0.1, 0.1, 0.1, 1.0]))
    assert_same_scene(got, cornell)
    assert got.num_triangles == 63978 and len(got.submeshes) == 11 and len(got.materials) == 5


def test_camera_matrices_match_the_python_convention():
    for (w, h) in [(512, 512), (1920, 1080)]:
        spec = host.camera_spec(position=(-0.025, 0.26, 0.75), target=(-0.025, 0.26, -0.3), fov_y=45.0, near=0.01, far=100.0)
        got = host.camera_matrices(spec, w, h).reshape(5, 16)
        want = camera.cornell_camera(w, h).reshape(5, 16)
        for k in range(5):
            scale = np.abs(want[k]).max()
            assert np.allclose(got[k], want[k], rtol=1e-5, atol=2e-6 * scale), k
    spec = host.camera_spec(position=(-12.5, 6.7, -1.85), pitch=0.208, yaw=1.731, fov_y=60.0, near=0.05, far=200.0)
    got = host.camera_matrices(spec, 1920, 1080).reshape(5, 16)
    want = camera.camera_buffer(camera.from_pitch_yaw((-12.5, 6.7, -1.85), 0.208, 1.731), camera.perspective(60.0, 1920 / 1080, 0.05, 200.0)).reshape(5, 16)
    assert np.allclose(got[3], want[3], rtol=1e-5, atol=1e-5) and np.allclose(got[4], want[4], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_layer_frame_sequencing(tmp_path):
    """RayTracingLayer.cpp:283,356-357,390-397,620: FrameIndex starts at 1, ++ per render, resets on resize,
    camera move, material change and when accumulation is off."""
    path = write_gltf(tmp_path)
    src = host.MeshSource(path)
    cam = host.camera_spec(position=(2.0, 3.0, 8.0), target=(2.0, 3.0, 3.0))
    layer = host.RayTracingLayer(src, camera=cam)
    layer.SetParams(Params.reference(spp_per_frame=2, max_bounces=3))
    layer.SetViewportSize(40, 24)
    assert layer.FrameIndex == 1
    for expect in (1, 2, 3):
        layer.OnUpdate()
        if expect == 1:
            assert layer.FrameIndex == 1         # the first Update sees the initial pose as "moved"
        layer.OnRender()
        assert layer.FrameIndex == expect + 1
    acc = layer.ReadAccumulation()
    assert (acc[..., 3] == 4).all()              # frames 2 and 3 accumulated 2 samples each; frame 1 only cleared
    img = layer.ReadImage()
    assert img[..., :3].max() > 0
    n = layer.RayCount()
    assert 40 * 24 * 2 * 3 <= n <= 40 * 24 * 2 * 3 * 3
    layer.SetCameraPosition((2.0, 3.0, 9.0))
    layer.OnUpdate()
    assert layer.FrameIndex == 1
    layer.OnRender()
    layer.SetViewportSize(32, 16)
    layer.OnUpdate()
    layer.OnRender()
    assert layer.FrameIndex == 2 and not layer.ReadAccumulation().any()
    layer.OnUpdate(); layer.OnRender()
    words = np.zeros(22, np.float32)
    words[0:3] = 0.9
    words[4] = 0.5
    words[8] = 1.0
    words.view(np.int32)[10:13] = -1
    words[21] = 1.5
    src.set_material(0, words)
    layer.MaterialsChanged()
    assert layer.FrameIndex == 1
    # a sky parameter change restarts the accumulation at the next update (m_UpdateSkyBox, RayTracingLayer.cpp:356-357)
    layer.OnUpdate(); layer.OnRender()
    layer.OnUpdate(); layer.OnRender()
    assert layer.FrameIndex == 3
    before = layer.ReadImage()
    layer.SetSkyboxSettings(2.0, 1.1, 0.7, use_as_environment=True, face_size=64)
    assert layer.FrameIndex == 3                 # dirty flag only; the reset happens in OnUpdate
    layer.OnUpdate()
    assert layer.FrameIndex == 1
    layer.OnRender()
    layer.OnUpdate()
    assert layer.FrameIndex == 2                 # the flag was cleared by the update that used it
    layer.OnRender()
    assert not np.array_equal(layer.ReadImage(), before)     # the paths now see the Preetham sky
    layer.SetAccumulate(False)
    layer.OnUpdate(); layer.OnRender()
    layer.OnUpdate()
    assert layer.FrameIndex == 1
    layer.close()


# ---- image files ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rle", [True, False])
@pytest.mark.parametrize("size", [(16, 8), (37, 5), (7, 3), (300, 2)])
def test_hdr_reader_matches_the_python_reader(tmp_path, rle, size):
    """No .hdr ships with the reference (its graveyard_pathways_4k.hdr is absent), so the files are made here."""
    from tests import hdr_py
    w, h = size
    rng = np.random.default_rng(w * 31 + h)
    rgb = np.exp(rng.uniform(-12, 12, (h, w, 3)))
    rgb[0, : w // 2] = rgb[0, 0]            # long runs for the run-length coder
    rgb[-1, 0] = 0.0
    rgbe = hdr_py.float_to_rgbe(rgb)
    path = tmp_path / "env.hdr"
    hdr_py.write_hdr(path, rgbe, rle=rle)
    got = host.load_hdr(path)
    want = hdr_py.read_hdr(path)
    assert got.shape == (h, w, 4) and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.all(got[..., 3] == 1.0) and np.all(got[-1, 0, :3] == 0.0)
    brightest = rgb.max(-1)
    lit = brightest > 0
    assert np.abs(got[..., :3].max(-1)[lit] / brightest[lit] - 1.0).max() < 0.01     # 8-bit mantissa of the brightest channel


def test_hdr_reader_errors(tmp_path):
    p = tmp_path / "bad.hdr"
    p.write_bytes(b"P6\n1 1\n255\n\x00\x00\x00")
    with pytest.raises(RuntimeError):
        host.load_hdr(p)
    p.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 4 +X 4\n\x01\x02")
    with pytest.raises(RuntimeError):
        host.load_hdr(p)
    with pytest.raises(RuntimeError):
        host.load_hdr(tmp_path / "missing.hdr")


def test_ppm_writer(tmp_path):
    img = np.arange(5 * 3 * 4, dtype=np.uint8).reshape(3, 5, 4)
    path = tmp_path / "out.ppm"
    host.write_ppm(path, img)
    data = path.read_bytes()
    assert data.startswith(b"P6\n5 3\n255\n")
    assert data[len(b"P6\n5 3\n255\n"):] == img[..., :3].tobytes()


@pytest.mark.gpu
def test_layer_post_processing_and_radiance_map(tmp_path):
    """OnRender ends with the post-processing pass (RayTracingLayer.cpp:421, exposure 0.8 by default); a radiance
    map loaded from a .hdr file restarts the accumulation and lights the scene."""
    from tests import hdr_py, oracle_lib
    path = write_gltf(tmp_path)
    src = host.MeshSource(path)
    cam = host.camera_spec(position=(2.0, 3.0, 8.0), target=(2.0, 3.0, 3.0))
    layer = host.RayTracingLayer(src, camera=cam)
    layer.SetParams(Params.reference(spp_per_frame=2, max_bounces=3))
    layer.SetViewportSize(40, 24)
    for _ in range(3):
        layer.OnUpdate(); layer.OnRender()
    img = layer.ReadImage()
    assert np.array_equal(layer.ReadPostProcessed(), oracle_lib.postprocess(img, 0.8))
    layer.SetPostProcessing(True, 2.0)
    layer.OnUpdate(); layer.OnRender()
    assert np.array_equal(layer.ReadPostProcessed(), oracle_lib.postprocess(layer.ReadImage(), 2.0))
    host.write_ppm(tmp_path / "frame.ppm", layer.ReadPostProcessed())
    assert (tmp_path / "frame.ppm").stat().st_size == len(b"P6\n40 24\n255\n") + 40 * 24 * 3
    # moving a submesh recreates the acceleration structure and changes the picture; FrameIndex is left alone
    before = layer.ReadImage()
    frame = layer.FrameIndex
    move = np.eye(4, dtype=np.float32)
    move[:3, 3] = (0.5, 0.25, 0.0)
    layer.SetSubMeshTransform(0, move)
    assert layer.FrameIndex == frame
    layer.SetAccumulate(False)
    layer.OnUpdate(); layer.OnRender()
    assert not np.array_equal(layer.ReadImage(), before)
    # dragging refits instead of rebuilding; the picture of a pose does not depend on how the hierarchy got there
    rebuilt = layer.ReadImage()
    move2 = move.copy()
    move2[:3, 3] = (-0.3, 0.1, 0.2)
    layer.SetSubMeshTransform(0, move2, refit=True)
    layer.SetSubMeshTransform(0, move, refit=True)
    layer.OnUpdate(); layer.OnRender()
    assert np.array_equal(layer.ReadImage(), rebuilt)
    layer.SetAccumulate(True)
    # a dark red sky: frame index restarts, and the sky colour shows in the pixels that miss everything
    sky = np.zeros((16, 32, 3))
    sky[..., 0] = 0.25
    hdr_py.write_hdr(tmp_path / "sky.hdr", hdr_py.float_to_rgbe(sky))
    layer.SetRadianceMap(tmp_path / "sky.hdr", 16)
    assert layer.FrameIndex == 1
    layer.OnUpdate(); layer.OnRender()
    img = layer.ReadImage()
    assert img[..., 0].max() > 0 and np.all(img[..., 1:3] <= img[..., 0:1] + 1e-6)
    layer.close()


# ---- texture decoders: checked against Pillow, an independent implementation ------------------------------------
def _pil():
    return pytest.importorskip("PIL.Image")


@pytest.mark.parametrize("mode,bits", [("L", 8), ("LA", 8), ("RGB", 8), ("RGBA", 8), ("P", 8), ("I;16", 16), ("1", 1)])
def test_png_decoder_matches_pillow(tmp_path, mode, bits):
    Image = _pil()
    rng = np.random.default_rng(len(mode) + bits)
    w, h = 37, 21
    if mode == "I;16":
        img = Image.fromarray(rng.integers(0, 65536, (h, w), dtype=np.uint16))
    elif mode == "1":
        img = Image.fromarray((rng.integers(0, 2, (h, w)) * 255).astype(np.uint8)).convert("1")
    elif mode == "P":
        img = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).quantize(17)
    else:
        ch = {"L": 1, "LA": 2, "RGB": 3, "RGBA": 4}[mode]
        a = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        a[:, : w // 2] = a[:, :1]                      # runs, so that the encoder picks different filters
        img = Image.fromarray(a[..., 0] if ch == 1 else a, mode)
    path = tmp_path / "t.png"
    img.save(path)
    got = host.load_image(path)
    if mode == "I;16":
        want = np.asarray(Image.open(path)).astype(np.uint16) >> 8
        want = np.stack([want, want, want, np.full_like(want, 255)], -1).astype(np.uint8)
    else:
        want = np.asarray(Image.open(path).convert("RGBA"))
    assert got.shape == (h, w, 4) and np.array_equal(got, want)


def test_png_decoder_errors(tmp_path):
    Image = _pil()
    p = tmp_path / "i.png"
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(p, interlace=True) if False else None
    raw = tmp_path / "raw.png"
    raw.write_bytes(b"\x89PNG\r\n\x1a\n" + b"\x00" * 20)
    with pytest.raises(RuntimeError):
        host.load_image(raw)
    raw.write_bytes(b"GIF89a")
    with pytest.raises(RuntimeError):
        host.load_image(raw)


@pytest.mark.parametrize("size,subsampling,quality,gray", [((64, 48), 0, 90, False), ((37, 29), 0, 75, False), ((40, 24), 2, 85, False),
                                                          ((33, 17), 1, 60, False), ((50, 50), 0, 95, True)])
def test_jpeg_decoder_matches_pillow(tmp_path, size, subsampling, quality, gray):
    """Baseline JPEG.  With 4:4:4 sampling (what all 65 Sponza textures use) the decoder follows the same published
    integer IDCT and colour conversion as libjpeg, so the bytes agree; with subsampled chroma Pillow interpolates
    where this decoder replicates, so only closeness is asked."""
    Image = _pil()
    w, h = size
    rng = np.random.default_rng(w * h)
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([127 + 120 * np.sin(xx / 5.0), 127 + 120 * np.cos(yy / 7.0), (xx * yy) % 256], -1) + rng.normal(0, 6, (h, w, 3))
    a = np.clip(a, 0, 255).astype(np.uint8)
    img = Image.fromarray(a[..., 0] if gray else a)
    path = tmp_path / "t.jpg"
    if gray:
        img.save(path, quality=quality)
    else:
        img.save(path, quality=quality, subsampling=subsampling)
    got = host.load_image(path).astype(np.int32)
    want = np.asarray(Image.open(path).convert("RGBA")).astype(np.int32)
    assert got.shape == want.shape and np.all(got[..., 3] == 255)
    diff = np.abs(got[..., :3] - want[..., :3])
    if subsampling == 0 or gray:
        assert diff.max() <= 1 and diff.mean() < 0.05
    else:
        assert diff.mean() < 12.0        # replicated against interpolated chroma, on a noisy image


def test_jpeg_restart_markers_and_errors(tmp_path):
    Image = _pil()
    a = (np.mgrid[0:32, 0:48].sum(0) * 3 % 256).astype(np.uint8)
    path = tmp_path / "r.jpg"
    Image.fromarray(np.stack([a, a.T[:32, :32].repeat(2, 1)[:, :48], 255 - a], -1)).save(path, quality=80, subsampling=0, restart_marker_blocks=2)
    got = host.load_image(path).astype(np.int32)
    want = np.asarray(Image.open(path).convert("RGBA")).astype(np.int32)
    assert np.abs(got - want).max() <= 1
    prog = tmp_path / "p.jpg"
    Image.fromarray(np.stack([a, a, a], -1)).save(prog, progressive=True)
    with pytest.raises(RuntimeError):
        host.load_image(prog)
    cut = tmp_path / "c.jpg"
    cut.write_bytes(path.read_bytes()[:200])
    with pytest.raises(RuntimeError):
        host.load_image(cut)


@pytest.mark.skipif(not have_ref, reason="reference assets not mounted")
def test_reference_textures_decode_like_pillow():
    """Suzanne's two PNGs through MeshSource.load_textures, and a sample of Sponza's PNG / JPEG files."""
    Image = _pil()
    src = host.MeshSource(os.path.join(REF_MODELS, "Suzanne/glTF/Suzanne.gltf"))
    assert src.load_textures() == 2
    sd = src.scene_data()
    for uri, tex in zip(src.texture_uris(), sd.textures):
        want = np.asarray(Image.open(os.path.join(REF_MODELS, "Suzanne/glTF", uri)).convert("RGBA"))
        assert np.array_equal(np.asarray(tex), want)
    import glob
    files = sorted(glob.glob(os.path.join(REF_MODELS, "Sponza/glTF/*.png")))[:2] + sorted(glob.glob(os.path.join(REF_MODELS, "Sponza/glTF/*.jpg")))[:4]
    for f in files:
        got = host.load_image(f).astype(np.int32)
        want = np.asarray(Image.open(f).convert("RGBA")).astype(np.int32)
        assert got.shape == want.shape and np.abs(got - want).max() <= 1, f


def test_parsers_survive_mutated_files(tmp_path):
    """tests/fuzz_parsers.cpp, built with the address and undefined-behaviour sanitizers, over corrupted copies of small
    PNG / JPEG / .hdr / glTF files: every input is either parsed or refused with an exception."""
    import shutil
    import subprocess
    from tests import hdr_py
    Image = pytest.importorskip("PIL.Image")
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hostdir = os.path.join(root, "pathtracer_amd", "host")
    exe = str(tmp_path / "fuzz_parsers")
    cmd = ["g++", "-O1", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I", hostdir,
           "-I", os.path.join(root, "include"), "-o", exe, os.path.join(root, "tests", "fuzz_parsers.cpp")]
    cmd += [os.path.join(hostdir, f) for f in ("gltf.cpp", "image_io.cpp", "image_decode.cpp")] + ["-lz"]
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0 and "sanitize" in built.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert built.returncode == 0, built.stderr[-2000:]
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (24, 20, 3), dtype=np.uint8)
    files = []
    for name, img, kw in (("a.png", Image.fromarray(a), {}), ("b.png", Image.fromarray(np.dstack([a, a[..., :1]])), {}),
                          ("c.png", Image.fromarray(a[..., 0]), {}), ("d.png", Image.fromarray(a).convert("P"), {}),
                          ("a.jpg", Image.fromarray(a), {"quality": 80, "subsampling": 0}),
                          ("b.jpg", Image.fromarray(a), {"quality": 60, "subsampling": 2}), ("c.jpg", Image.fromarray(a[..., 0]), {})):
        img.save(tmp_path / name, **kw)
        files.append(str(tmp_path / name))
    hdr_py.write_hdr(tmp_path / "a.hdr", hdr_py.float_to_rgbe(rng.uniform(0, 4, (16, 32, 3))))
    files += [str(tmp_path / "a.hdr"), write_gltf(tmp_path)] + sorted(hostile_gltfs(tmp_path).values())
    env = dict(os.environ, ITERS="400", SEED="7", ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0")
    run = subprocess.run([exe, str(tmp_path)] + files, capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout + run.stderr)[-2000:]
    assert "parsed" in run.stdout
