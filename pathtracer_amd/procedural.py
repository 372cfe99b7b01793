"""Seeded procedural stand-ins for the scenes whose geometry is not in the reference checkout.

`Sponza.bin` (9.5 MB of geometry behind PathTracer/assets/models/Sponza/glTF/Sponza.gltf: 1 mesh,
103 primitives, 25 materials, 262 267 triangles, node scale 0.008) and the Bistro scene are missing
from the reference repository, so throughput is measured on synthetic scenes of the same class:

  sponza_class()  an arcaded two-storey atrium, open to the sky above the nave, with columns, arches,
                  curtains, banners, vases, plants and lamps: ~262 k triangles, 103 submeshes,
                  25 materials, every submesh under the same 0.008 node scale as Sponza.
  bistro_class()  a street block with facades, props and leafy trees, ~3 M triangles.

Everything is generated with numpy from a fixed seed; the same arrays go to the GPU backend and to
the CPU oracle, so parity does not depend on how the generator rounds.
"""
import math

import numpy as np

from .scene import SceneData, make_material


class _Builder:
    def __init__(self, scale):
        self.v, self.i, self.sub, self.cnt, self.xf, self.names = [], [], [], [], [], []
        self.nv = self.ni = 0
        self.scale = scale

    def add(self, name, pos, nrm, uv, tris, material):
        pos = np.asarray(pos, np.float64).reshape(-1, 3)
        n = len(pos)
        v = np.zeros((n, 12), np.float32)
        v[:, 0:3] = pos / self.scale          # object space; the submesh transform scales it back
        v[:, 3:5] = uv
        nn = np.asarray(nrm, np.float64).reshape(-1, 3)
        nn = nn / np.maximum(np.linalg.norm(nn, axis=1, keepdims=True), 1e-30)
        v[:, 5:8] = nn
        v[:, 8:12] = (1, 0, 0, 1)
        tris = np.asarray(tris, np.uint32).reshape(-1)
        self.v.append(v)
        self.i.append(tris)
        self.sub.append((0, self.nv, self.ni, material))
        self.cnt.append(len(tris))
        s = self.scale
        self.xf.append(np.array([s, 0, 0, 0, 0, s, 0, 0, 0, 0, s, 0], np.float32))
        self.names.append(name)
        self.nv += n
        self.ni += len(tris)

    def merge_add(self, name, parts, material):
        pos, nrm, uv, tris, base = [], [], [], [], 0
        for (p, n, u, t) in parts:
            pos.append(p); nrm.append(n); uv.append(u); tris.append(np.asarray(t, np.int64) + base)
            base += len(p)
        self.add(name, np.concatenate(pos), np.concatenate(nrm), np.concatenate(uv), np.concatenate(tris), material)

    def scene(self, materials):
        return SceneData(np.concatenate(self.v), np.concatenate(self.i), np.array(self.sub, np.uint32),
                         np.array(self.cnt, np.uint32), np.array(self.xf, np.float32), np.array(materials, np.float32),
                         names=self.names)


def _grid(origin, du, dv, nu, nv, normal=None, disp=None):
    """(nu x nv) quads spanning origin + s*du + t*dv; disp(s, t) -> offset along the normal."""
    s, t = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    o, du, dv = np.asarray(origin, float), np.asarray(du, float), np.asarray(dv, float)
    p = o + s[..., None] * du + t[..., None] * dv
    n = np.cross(du, dv) if normal is None else np.asarray(normal, float)
    n = n / np.linalg.norm(n)
    if disp is not None:
        p = p + disp(s, t)[..., None] * n
        # normals from finite differences
        ds = np.gradient(p, axis=0)
        dt = np.gradient(p, axis=1)
        nn = np.cross(ds, dt)
    else:
        nn = np.broadcast_to(n, p.shape)
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    tris = np.stack([a, b, c, a, c, d], axis=-1).reshape(-1)
    uv = np.stack([s * nu * 0.5, t * nv * 0.5], axis=-1)
    return p.reshape(-1, 3), np.asarray(nn).reshape(-1, 3), uv.reshape(-1, 2), tris


def _box(lo, hi, n=1):
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    e = hi - lo
    faces = [
        (lo, (e[0], 0, 0), (0, 0, e[2]), (0, -1, 0)), ((lo[0], hi[1], lo[2]), (0, 0, e[2]), (e[0], 0, 0), (0, 1, 0)),
        (lo, (0, e[1], 0), (e[0], 0, 0), (0, 0, -1)), ((lo[0], lo[1], hi[2]), (e[0], 0, 0), (0, e[1], 0), (0, 0, 1)),
        (lo, (0, 0, e[2]), (0, e[1], 0), (-1, 0, 0)), ((hi[0], lo[1], lo[2]), (0, e[1], 0), (0, 0, e[2]), (1, 0, 0)),
    ]
    return [_grid(o, du, dv, n, n, normal=nrm) for (o, du, dv, nrm) in faces]


def _revolve(profile, center, seg):
    """Surface of revolution about +Y: profile = [(radius, y), ...]."""
    prof = np.asarray(profile, float)
    ang = np.linspace(0, 2 * math.pi, seg + 1)
    r, y = prof[:, 0][:, None], prof[:, 1][:, None]
    p = np.stack([r * np.cos(ang), np.broadcast_to(y, (len(prof), seg + 1)), r * np.sin(ang)], axis=-1) + np.asarray(center, float)
    dr = np.gradient(prof[:, 0])
    dy = np.gradient(prof[:, 1])
    nr, ny = dy[:, None], -dr[:, None]
    n = np.stack([nr * np.cos(ang), np.broadcast_to(ny, (len(prof), seg + 1)), nr * np.sin(ang)], axis=-1)
    idx = np.arange(len(prof) * (seg + 1)).reshape(len(prof), seg + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    tris = np.stack([a, c, b, a, d, c], axis=-1).reshape(-1)
    uv = np.stack(np.meshgrid(np.linspace(0, 4, len(prof)), np.linspace(0, 8, seg + 1), indexing="ij"), axis=-1)
    return p.reshape(-1, 3), n.reshape(-1, 3), uv.reshape(-1, 2), tris


def _sphere(center, radius, rings, seg):
    th = np.linspace(0, math.pi, rings + 1)
    prof = np.stack([radius * np.sin(th), -radius * np.cos(th)], axis=1)
    p, n, uv, t = _revolve(prof, center, seg)
    n = p - np.asarray(center, float)
    return p, n, uv, t


def _arch(p0, p1, y0, rise, depth, seg):
    """Semi-circular arch band between two column tops, extruded by `depth` across."""
    p0, p1 = np.asarray(p0, float), np.asarray(p1, float)
    span = p1 - p0
    length = np.linalg.norm(span)
    ax = span / length
    across = np.cross(ax, (0, 1, 0))
    ang = np.linspace(0, math.pi, seg + 1)
    mid = (p0 + p1) * 0.5
    pts = mid[None, :] - np.cos(ang)[:, None] * ax * (length * 0.5) + np.sin(ang)[:, None] * np.array([0, rise, 0]) + np.array([0, y0, 0])
    parts = []
    for side in (-0.5, 0.5):
        a = pts + across * depth * side
        top = a.copy()
        top[:, 1] = y0 + rise * 1.25
        p = np.concatenate([a, top])
        n = np.broadcast_to(across * (1 if side > 0 else -1), p.shape)
        idx = np.arange(2 * (seg + 1)).reshape(2, seg + 1)
        q0, q1, q2, q3 = idx[0, :-1], idx[0, 1:], idx[1, 1:], idx[1, :-1]
        t = np.stack([q0, q1, q2, q0, q2, q3], axis=-1).reshape(-1)
        uv = np.stack([np.tile(np.linspace(0, 2, seg + 1), 2), np.repeat([0.0, 1.0], seg + 1)], axis=1)
        parts.append((p, n, uv, t))
    # soffit (underside of the arch)
    a = pts - across * depth * 0.5
    b = pts + across * depth * 0.5
    p = np.concatenate([a, b])
    nrm = np.concatenate([mid + [0, y0, 0] - a, mid + [0, y0, 0] - b])
    idx = np.arange(2 * (seg + 1)).reshape(2, seg + 1)
    q0, q1, q2, q3 = idx[0, :-1], idx[0, 1:], idx[1, 1:], idx[1, :-1]
    t = np.stack([q0, q2, q1, q0, q3, q2], axis=-1).reshape(-1)
    uv = np.stack([np.tile(np.linspace(0, 2, seg + 1), 2), np.repeat([0.0, 1.0], seg + 1)], axis=1)
    parts.append((p, nrm, uv, t))
    return parts


def _leaf_cards(rng, center, radius, count, size):
    c = np.asarray(center, float) + rng.normal(size=(count, 3)) * radius * np.array([1, 0.7, 1])
    u = rng.normal(size=(count, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    w = np.cross(u, rng.normal(size=(count, 3))); w /= np.linalg.norm(w, axis=1, keepdims=True)
    s = size * rng.uniform(0.6, 1.4, (count, 1))
    p = np.stack([c - u * s, c + w * s * 0.6, c + u * s, c - w * s * 0.6], axis=1).reshape(-1, 3)
    n = np.repeat(np.cross(u, w), 4, axis=0)
    base = np.arange(count)[:, None] * 4
    t = (base + np.array([0, 1, 2, 0, 2, 3])).reshape(-1)
    uv = np.tile(np.array([[0, 0], [1, 0], [1, 1], [0, 1]], float), (count, 1))
    return p, n, uv, t


def sponza_materials(full_disney=True):
    """25 materials.  With full_disney a fixed third each gets clearcoat / sheen / specular transmission."""
    rng = np.random.default_rng(7)
    mats = []
    base = [(0.75, 0.71, 0.62), (0.62, 0.55, 0.45), (0.8, 0.78, 0.74), (0.55, 0.33, 0.25), (0.35, 0.42, 0.3),
            (0.7, 0.12, 0.1), (0.12, 0.3, 0.62), (0.1, 0.5, 0.2), (0.85, 0.75, 0.3), (0.9, 0.9, 0.9)]
    for k in range(25):
        col = base[k % len(base)]
        col = tuple(float(np.clip(c * rng.uniform(0.85, 1.1), 0.03, 0.95)) for c in col)
        kw = dict(albedo=col, metallic=0.0, roughness=float(rng.uniform(0.35, 0.95)), ior=1.5)
        if k in (8, 18):
            kw.update(metallic=1.0, roughness=float(rng.uniform(0.15, 0.4)))       # brass, steel
        if full_disney:
            if k % 3 == 0:
                kw.update(clearcoat=1.0, clearcoat_roughness=0.1)
            elif k % 3 == 1:
                kw.update(sheen=1.0, sheen_tint=0.5)
            elif k not in (2, 8):
                kw.update(spec_trans=1.0, ior=1.5, roughness=float(rng.uniform(0.05, 0.3)))
        mats.append(make_material(**kw))
    mats[24] = make_material(albedo=(1, 1, 1), roughness=1.0, emissive=(1.0, 0.85, 0.6), emissive_strength=12.0)  # lamps
    return mats


def sponza_textures(size=512, seed=11):
    """Twelve RGBA8 base-colour textures (bricks, tiles, weave, marble veins, wood grain, plaster ...) and four
    metallic-roughness maps (G = roughness factor, B = metallic factor, ClosestHit.glsl:182), generated from integer
    hashes and piecewise-linear waves only, so that every machine produces the same bytes."""
    n = size
    y, x = np.mgrid[0:n, 0:n].astype(np.uint32)

    def hash2(a, b, k):
        h = (a * np.uint32(0x9E3779B1) + b * np.uint32(0x85EBCA77) + np.uint32((k * 0xC2B2AE3D) & 0xFFFFFFFF)).astype(np.uint32)
        h ^= h >> np.uint32(15)
        h = (h * np.uint32(0x2C1B3C6D)).astype(np.uint32)
        h ^= h >> np.uint32(12)
        return (h & np.uint32(0xFFFF)).astype(np.float32) / np.float32(65535.0)

    def value_noise(cells, k):
        c = n // cells
        gx, gy = x // c, y // c
        fx, fy = (x % c).astype(np.float32) / c, (y % c).astype(np.float32) / c
        v = lambda ix, iy: hash2(ix % cells, iy % cells, k)   # noqa: E731
        top = v(gx, gy) * (1 - fx) + v(gx + 1, gy) * fx
        bot = v(gx, gy + 1) * (1 - fx) + v(gx + 1, gy + 1) * fx
        return top * (1 - fy) + bot * fy

    def tri(t):                                  # triangle wave in [0, 1], period 1
        f = t - np.floor(t)
        return 1 - np.abs(2 * f - 1)

    u, v = x.astype(np.float32) / n, y.astype(np.float32) / n
    fine = value_noise(64, seed) * 0.6 + value_noise(128, seed + 1) * 0.4
    coarse = value_noise(8, seed + 2)
    brick = np.minimum(((v * 8 + 0.02) % 1.0 > 0.08), (((u * 4 + 0.5 * (np.floor(v * 8) % 2)) % 1.0) > 0.04)).astype(np.float32)
    tiles = (((u * 6) % 1.0 > 0.03) & ((v * 6) % 1.0 > 0.03)).astype(np.float32)
    weave = 0.5 + 0.5 * tri(u * 48) * tri(v * 48 + 0.25)
    veins = np.clip(1.0 - 6.0 * np.abs(tri(u * 3 + coarse * 0.7 + v) - 0.5), 0, 1)
    grain = tri(v * 40 + coarse * 3.0)
    stripes = (tri(u * 10) > 0.5).astype(np.float32)
    patterns = [0.55 + 0.45 * brick * (0.8 + 0.2 * fine), 0.5 + 0.5 * tiles * (0.85 + 0.15 * coarse), weave, 1.0 - 0.6 * veins,
                0.6 + 0.4 * grain, 0.7 + 0.3 * fine, 0.6 + 0.4 * stripes, 0.5 + 0.5 * coarse, 0.75 + 0.25 * tri(u * 24 + fine),
                0.65 + 0.35 * (fine > 0.5), 0.8 + 0.2 * tri((u + v) * 16), 0.55 + 0.45 * tiles * weave]
    tints = [(1.0, 0.93, 0.86), (0.95, 0.95, 1.0), (1.0, 0.85, 0.85), (0.97, 0.97, 0.95), (1.0, 0.9, 0.75), (0.96, 0.96, 0.96),
             (0.9, 0.95, 1.0), (0.9, 1.0, 0.9), (1.0, 0.97, 0.8), (0.93, 0.9, 1.0), (1.0, 0.92, 0.92), (0.92, 1.0, 0.98)]
    out = []
    for p_, t_ in zip(patterns, tints):
        rgb = np.stack([np.clip(p_ * c, 0, 1) for c in t_], axis=-1)
        out.append(np.concatenate([np.round(rgb * 255).astype(np.uint8), np.full((n, n, 1), 255, np.uint8)], axis=-1))
    for k in range(4):
        rough = np.clip(0.55 + 0.45 * (fine if k & 1 else coarse) * (0.7 + 0.3 * tiles), 0, 1)
        metal = np.clip(0.85 + 0.15 * value_noise(16, seed + 5 + k), 0, 1)
        mr = np.stack([np.zeros_like(rough), rough, metal, np.ones_like(rough)], axis=-1)
        out.append(np.round(mr * 255).astype(np.uint8))
    return out


def sponza_class(seed=1, full_disney=True, detail=1.0, layout="uniform", textures=False):
    """~262 k triangles / 103 submeshes / 25 materials at detail = 1.
    layout "uniform": every surface finely tessellated (largest triangle 9 % of the scene diagonal).
    layout "sponza": the size distribution of an architectural model like the real Sponza: floor, walls, gallery slabs and
    roof as a few large quads (triangles up to a third of the scene), long thin trims and rails (aspect up to 300 : 1) next to
    the finely tessellated columns, cloth, vases and plants.
    textures: base-colour maps on every material but the lamps, metallic-roughness maps on a third (Sponza has 69 textures)."""
    rng = np.random.default_rng(seed)
    b = _Builder(0.008)
    arch = layout == "sponza"
    assert layout in ("uniform", "sponza")
    boost = 1.04 if arch else 1.0                  # the fine parts take over the triangles the large surfaces give up
    d = lambda n: max(2, int(round(n * math.sqrt(detail) * boost)))   # noqa: E731  tessellation per axis
    coarse = lambda n, few: few if arch else d(n) if boost == 1.0 else max(2, int(round(n * math.sqrt(detail))))   # noqa: E731
    X, Z, H1, H2 = 15.0, 9.0, 5.5, 11.0            # half length, half width, storey heights
    nave = 4.2                                       # half width of the open nave
    # shell: floor, four walls (two storeys each), gallery floors, roof strips around the open nave
    b.merge_add("floor", [_grid((-X, 0, -Z), (0, 0, 2 * Z), (2 * X, 0, 0), coarse(40, 4), coarse(64, 6),
                                disp=None if arch else (lambda s, t: 0.01 * np.sin(40 * s) * np.sin(60 * t)))], 0)
    walls = [((-X, 0, -Z), (2 * X, 0, 0), (0, H2 + 1, 0)), ((X, 0, Z), (-2 * X, 0, 0), (0, H2 + 1, 0)),
             ((-X, 0, Z), (0, 0, -2 * Z), (0, H2 + 1, 0)), ((X, 0, -Z), (0, 0, 2 * Z), (0, H2 + 1, 0))]
    for k, (o, du, dv) in enumerate(walls):
        parts = [_grid(o, du, dv, coarse(48, 3), coarse(24, 2), disp=None if arch else (lambda s, t: 0.03 * np.sin(24 * math.pi * s) * np.sin(12 * math.pi * t)))]
        if arch:
            # cornices and skirtings: boxes as long as the wall, a hand wide
            o_, du_ = np.asarray(o, float), np.asarray(du, float)
            inward = np.cross(du_ / np.linalg.norm(du_), (0, 1, 0))
            for y in (0.0, 1.1, H1 - 0.6, H1 + 0.1, H2 - 0.4, H2 + 0.7):
                p0 = o_ + inward * 0.0 + np.array([0, y, 0])
                p1 = o_ + du_ + inward * 0.12 + np.array([0, y + 0.1, 0])
                parts += _box(np.minimum(p0, p1), np.maximum(p0, p1), 1)
        b.merge_add("wall%d" % k, parts, 1 + (k & 1))
    for k, z0 in enumerate((-Z, nave)):
        b.merge_add("gallery%d" % k, _box((-X, H1 - 0.25, z0), (X, H1, z0 + (Z - nave)), coarse(10, 1)), 3)
        b.merge_add("roof%d" % k, _box((-X, H2, z0), (X, H2 + 0.3, z0 + (Z - nave)), coarse(10, 1)), 3)
    ends = [_box((x0, H2, -nave), (x0 + 3.0, H2 + 0.3, nave), coarse(6, 1)) for x0 in (-X, X - 3.0)]
    if arch:
        b.merge_add("roof_ends", ends[0] + ends[1], 3)     # one submesh, so that the beams below keep the count at 103
    else:
        for k in range(2):
            b.merge_add("roof_end%d" % k, ends[k], 3)
    # columns: 2 rows x 11 x 2 storeys, each its own submesh
    xs = np.linspace(-X + 2.0, X - 2.0, 11)
    ci = 0
    for storey, (y0, y1) in enumerate(((0.0, H1 - 0.25), (H1, H2))):
        for row, z in enumerate((-nave, nave)):
            for x in xs:
                h = y1 - y0
                prof = [(0.55, 0), (0.55, 0.25), (0.42, 0.32), (0.38, 0.5)] + \
                       [(0.38 - 0.05 * t, 0.5 + (h - 1.2) * t) for t in np.linspace(0, 1, d(14))[1:]] + \
                       [(0.36, h - 0.62), (0.5, h - 0.45), (0.58, h - 0.3), (0.58, h)]
                prof = [(r, y + y0) for r, y in prof]
                flutes = _revolve(prof, (x, 0, z), d(28))
                p = flutes[0]
                ang = np.arctan2(p[:, 2] - z, p[:, 0] - x)
                rad = 1.0 + 0.025 * np.cos(16 * ang)
                p2 = p.copy()
                p2[:, 0] = x + (p[:, 0] - x) * rad
                p2[:, 2] = z + (p[:, 2] - z) * rad
                b.merge_add("column%d" % ci, [(p2, flutes[1], flutes[2], flutes[3])], 4 + (storey * 2 + row) % 3)
                ci += 1
    # arches between neighbouring columns, grouped per row and storey
    for storey, ytop in enumerate((H1 - 0.25 - 1.3, H2 - 1.3)):
        for row, z in enumerate((-nave, nave)):
            parts = []
            for x0, x1 in zip(xs[:-1], xs[1:]):
                parts += _arch((x0 + 0.4, 0, z), (x1 - 0.4, 0, z), ytop, 1.0, 0.7, d(20))
            b.merge_add("arches%d_%d" % (storey, row), parts, 7)
    # curtains and banners: dense wavy cloth
    for k in range(8):
        x = xs[1 + k] + 0.2
        z = nave if k & 1 else -nave
        ph = rng.uniform(0, 6)
        b.merge_add("curtain%d" % k, [_grid((x, H1 - 0.4, z), (2.3, 0, 0), (0, -3.6, 0), d(58), d(58),
                                            disp=lambda s, t, ph=ph: 0.12 * np.sin(22 * s + ph) * (0.3 + t))], 9 + k % 3)
    for k in range(6):
        x = -10.0 + 4.0 * k
        ph = rng.uniform(0, 6)
        b.merge_add("banner%d" % k, [_grid((x, H2 - 0.5, -1.2), (0, 0, 2.4), (0, -4.5, 0), d(40), d(70),
                                           disp=lambda s, t, ph=ph: 0.2 * np.sin(9 * s + 5 * t + ph) * t)], 12 + k % 3)
    # vases, plants, spheres
    for k in range(10):
        x = -12.0 + 2.6 * k + rng.uniform(-0.3, 0.3)
        z = (-1) ** k * (nave - 1.2)
        prof = [(0.02, 0.0), (0.3, 0.0), (0.38, 0.1), (0.52, 0.45), (0.42, 0.85), (0.22, 1.05), (0.3, 1.2), (0.27, 1.2), (0.18, 1.05)]
        prof = np.array(prof)
        tt = np.linspace(0, len(prof) - 1, d(34))
        prof = np.stack([np.interp(tt, np.arange(len(prof)), prof[:, 0]), np.interp(tt, np.arange(len(prof)), prof[:, 1])], axis=1)
        b.merge_add("vase%d" % k, [_revolve(prof, (x, 0, z), d(40))], 15 + k % 3)
    for k in range(6):
        x = -11.0 + 4.4 * k
        z = (-1) ** (k + 1) * (nave - 1.2)
        b.merge_add("plant%d" % k, [_leaf_cards(rng, (x, 1.9, z), 0.55, int(4000 * detail), 0.1)], 18 + k % 2)
    for k in range(6):
        x = -10.0 + 4.0 * k
        b.merge_add("orb%d" % k, [_sphere((x, 0.6 + 0.25 * (k % 3), 0.0 + 1.5 * (-1) ** k), 0.6, d(32), d(64))], 20 + k % 4)
    for k in range(4):
        x = -9.0 + 6.0 * k
        b.merge_add("bench%d" % k, _box((x, 0.0, -0.35), (x + 2.2, 0.5, 0.35), coarse(6, 1)), 5)
    # lamps: small emissive spheres under the galleries
    for k, z in enumerate((-Z + 2.0, Z - 2.0)):
        parts = [_sphere((x, H1 - 0.9, z), 0.22, d(10), d(16)) for x in xs[::2]]
        b.merge_add("lamps%d" % k, parts, 24)
    # upper balustrade rails
    for k, z in enumerate((-nave + 0.2, nave - 0.2)):
        parts = []
        for x in np.linspace(-X + 2.2, X - 2.2, int(60 * detail) + 2):
            parts.append(_revolve([(0.05, H1), (0.09, H1 + 0.3), (0.05, H1 + 0.6), (0.08, H1 + 0.9)], (x, 0, z), 8))
        parts += _box((-X + 2.0, H1 + 0.9, z - 0.1), (X - 2.0, H1 + 1.0, z + 0.1), coarse(8, 1))
        b.merge_add("balustrade%d" % k, parts, 3)
    if arch:
        # what hurts a hierarchy most: long thin parts that run diagonally through the volume, so that their boxes are
        # almost empty.  Tie beams from the gallery on one side to the roof edge on the other, skewed along the nave, and
        # tilted louvre slats over the open nave (sky still shows between them).
        def oriented_box(p0, p1, half_w, half_h):
            p0, p1 = np.asarray(p0, float), np.asarray(p1, float)
            ax = p1 - p0
            side = np.cross(ax, (0, 1, 0)); side /= np.linalg.norm(side)
            up = np.cross(side, ax); up /= np.linalg.norm(up)
            parts = []
            for sgn_s, sgn_u, (du_, dv_) in ((0, -1, (side * 2 * half_w, ax)), (0, 1, (ax, side * 2 * half_w)),
                                             (-1, 0, (ax, up * 2 * half_h)), (1, 0, (up * 2 * half_h, ax))):
                o = p0 - side * half_w - up * half_h
                if sgn_u > 0:
                    o = o + up * 2 * half_h
                if sgn_s > 0:
                    o = o + side * 2 * half_w
                parts.append(_grid(o, du_, dv_, 1, 1))
            return parts
        beams = []
        for k, x in enumerate(np.linspace(-X + 3.0, X - 7.0, 12)):
            z0, z1 = (-nave, nave) if k % 2 == 0 else (nave, -nave)
            beams += oriented_box((x, H1 + 0.8, z0), (x + 4.0, H2 - 0.4, z1), 0.09, 0.12)
        slats = []
        for x in np.linspace(-X + 2.0, X - 4.5, 9):
            slats.append(_grid((x, H2 + 0.2, -nave - 0.3), (1.9, 1.3, 0.0), (0.0, 0.0, 2 * nave + 0.6), 1, 1))
        b.merge_add("beams_and_louvres", beams + slats, 7)
    mats = sponza_materials(full_disney)
    sc = b.scene(mats)
    if textures:
        sc.textures = sponza_textures()
        sc.texture_flags = [0] * len(sc.textures)
        ints = sc.materials.view(np.int32)
        for k in range(24):                      # not the lamps
            ints[k, 10] = k % 12                  # base colour
            if k % 3 == 1:
                ints[k, 11] = 12 + (k // 3) % 4   # metallic-roughness
    return sc


def sponza_camera(width, height):
    """The reference's commented Sponza pose (RayTracingLayer.cpp:39-42: pitch 0.208, yaw 1.731,
    position (-12.5, 6.7, -1.85)) as explicit matrices."""
    from . import camera
    view = camera.from_pitch_yaw((-12.5, 6.7, -1.85), 0.208, 1.731)
    return camera.camera_buffer(view, camera.perspective(60.0, width / height, 0.05, 200.0))


def bistro_class(seed=2, target_tris=3_000_000):
    """A street block: ground, facades with window insets, props and trees; about `target_tris` triangles."""
    rng = np.random.default_rng(seed)
    b = _Builder(1.0)
    mats = sponza_materials(True)
    L = 60.0
    b.merge_add("street", [_grid((-L, 0, -12), (0, 0, 24), (2 * L, 0, 0), 120, 600,
                                 disp=lambda s, t: 0.02 * np.sin(90 * s) * np.sin(300 * t))], 0)
    k = 0
    for side in (-1, 1):
        x = -L
        while x < L - 4:
            w = rng.uniform(8, 14)
            h = rng.uniform(9, 18)
            z0 = side * 12.0
            parts = _box((x, 0, min(z0, z0 + side * 10)), (x + w, h, max(z0, z0 + side * 10)), 24)
            # window boxes on the street-facing facade
            for wy in np.arange(1.5, h - 2.0, 3.0):
                for wx in np.arange(x + 1.0, x + w - 1.5, 2.5):
                    zf = z0 - side * 0.15
                    parts += _box((wx, wy, min(zf, z0)), (wx + 1.4, wy + 1.8, max(zf, z0)), 3)
            b.merge_add("building%d" % k, parts, 1 + k % 8)
            x += w + rng.uniform(0.0, 1.5)
            k += 1
    used = sum(b.cnt) // 3
    trees = 36
    per_tree = max(1000, (target_tris - used - 400_000) // (trees * 2))
    for t in range(trees):
        x = -L + 4 + (2 * L - 8) * (t // 2) / (trees // 2 - 1)
        z = 8.5 * (-1) ** t
        trunk = _revolve([(0.35, 0), (0.25, 2.0), (0.18, 4.0), (0.05, 5.5)], (x, 0, z), 24)
        b.merge_add("trunk%d" % t, [trunk], 3)
        b.merge_add("leaves%d" % t, [_leaf_cards(rng, (x, 6.0, z), 1.8, int(per_tree), 0.12)], 18 + t % 2)
    used = sum(b.cnt) // 3
    props = 160
    per = max(8, int(math.sqrt(max(0, target_tris - used) / props / 4)))
    for p in range(props):
        x, z = rng.uniform(-L + 2, L - 2), rng.uniform(-7.5, 7.5)
        if p % 2:
            b.merge_add("prop%d" % p, [_sphere((x, 0.5, z), 0.5, per, 2 * per)], 9 + p % 14)
        else:
            prof = np.array([(0.02, 0.0), (0.3, 0.0), (0.4, 0.4), (0.25, 0.9), (0.3, 1.1)])
            tt = np.linspace(0, len(prof) - 1, per)
            prof = np.stack([np.interp(tt, np.arange(len(prof)), prof[:, 0]), np.interp(tt, np.arange(len(prof)), prof[:, 1])], axis=1)
            b.merge_add("prop%d" % p, [_revolve(prof, (x, 0, z), 2 * per)], 9 + p % 14)
    parts = [_sphere((x, 7.0, 0.0), 0.3, 10, 16) for x in np.linspace(-L + 5, L - 5, 12)]
    b.merge_add("street_lamps", parts, 24)
    return b.scene(mats)


def bistro_camera(width, height):
    from . import camera
    view = camera.look_at((-50.0, 3.2, 1.5), (0.0, 5.0, -1.0))
    return camera.camera_buffer(view, camera.perspective(60.0, width / height, 0.05, 500.0))


def sky_mirrored_below_the_horizon(cube):
    """The reference's Preetham sky pass (PreethamSky.glsl) only has values above the horizon: its Perez term divides by
    cos(theta), which is 0 on the horizon and clamped there below it, so the lower half of the cube is NaN or infinite, and one
    such texel under a path turns its pixel NaN for good.  For a workload whose rays can leave downwards or along the horizon,
    this fills every non-finite texel with the sky's value in the direction mirrored at the horizon plane (y -> -y).
    cube: (6, n, n, 4), faces +X -X +Y -Y +Z -Z with rows running along -y on the four side faces; returns a new array."""
    cube = np.array(cube, np.float32, copy=True)
    for f in (0, 1, 4, 5):
        bad = ~np.isfinite(cube[f]).all(-1)
        cube[f][bad] = cube[f][::-1][bad]
    bad = ~np.isfinite(cube[3]).all(-1)
    cube[3][bad] = cube[2][::-1][bad]        # -Y from +Y: same s = x, t = -z against t = z
    # what is still not finite (the horizon row itself, if any) takes the nearest finite row above it
    for f in (0, 1, 4, 5):
        for _ in range(4):
            bad = ~np.isfinite(cube[f]).all(-1)
            if not bad.any():
                break
            up = np.roll(cube[f], 1, axis=0)
            cube[f][bad] = up[bad]
    cube[~np.isfinite(cube).all(-1)] = (0.0, 0.0, 0.0, 1.0)
    return cube
