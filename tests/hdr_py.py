"""Independent Python reader and writer of Radiance .hdr (RGBE) files, used to check the C++ reader of the host
layer.  Written from the format description (G. Ward, "Real Pixels", Graphics Gems II), not from the C++ code."""
import numpy as np


def float_to_rgbe(rgb):
    rgb = np.asarray(rgb, np.float64)
    m = rgb.max(axis=-1)
    out = np.zeros(rgb.shape[:-1] + (4,), np.uint8)
    nz = m > 1e-32
    mant, exp = np.frexp(m[nz])
    scale = mant * 256.0 / m[nz]
    out[nz, :3] = np.clip(rgb[nz] * scale[:, None], 0, 255).astype(np.uint8)
    out[nz, 3] = (exp + 128).astype(np.uint8)
    return out


def rgbe_to_float(rgbe):
    rgbe = np.asarray(rgbe, np.uint8)
    e = rgbe[..., 3].astype(np.int32)
    f = np.where(e > 0, np.ldexp(1.0, e - 136), 0.0)
    out = np.ones(rgbe.shape[:-1] + (4,), np.float32)
    out[..., :3] = (rgbe[..., :3].astype(np.float64) * f[..., None]).astype(np.float32)
    return out


def _rle_channel(row):
    out = bytearray()
    i, n = 0, len(row)
    while i < n:
        run = 1
        while i + run < n and run < 127 and row[i + run] == row[i]:
            run += 1
        if run >= 4:
            out += bytes([128 + run, row[i]])
            i += run
        else:
            j = i
            lit = bytearray()
            while j < n and len(lit) < 128:
                r = 1
                while j + r < n and r < 4 and row[j + r] == row[j]:
                    r += 1
                if r >= 4:
                    break
                lit.append(row[j])
                j += 1
            out += bytes([len(lit)]) + bytes(lit)
            i = j
    return bytes(out)


def write_hdr(path, rgbe, rle=True, header_extra=b"EXPOSURE=1.0\n"):
    h, w = rgbe.shape[:2]
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\n# made by tests/hdr_py.py\nFORMAT=32-bit_rle_rgbe\n" + header_extra + b"\n")
        f.write(b"-Y %d +X %d\n" % (h, w))
        for y in range(h):
            if rle and 8 <= w < 32768:
                f.write(bytes([2, 2, w >> 8, w & 255]))
                for ch in range(4):
                    f.write(_rle_channel(rgbe[y, :, ch].tolist()))
            else:
                f.write(rgbe[y].tobytes())


def read_hdr(path):
    data = open(path, "rb").read()
    end = data.index(b"\n\n")
    header = data[:end].split(b"\n")
    assert header[0] in (b"#?RADIANCE", b"#?RGBE") and b"FORMAT=32-bit_rle_rgbe" in header
    pos = end + 2
    eol = data.index(b"\n", pos)
    parts = data[pos:eol].split()
    assert parts[0] == b"-Y" and parts[2] == b"+X"
    h, w = int(parts[1]), int(parts[3])
    pos = eol + 1
    rgbe = np.zeros((h, w, 4), np.uint8)
    for y in range(h):
        if 8 <= w < 32768 and data[pos] == 2 and data[pos + 1] == 2 and (data[pos + 2] << 8 | data[pos + 3]) == w:
            pos += 4
            for ch in range(4):
                x = 0
                while x < w:
                    c = data[pos]; pos += 1
                    if c > 128:
                        rgbe[y, x:x + c - 128, ch] = data[pos]; pos += 1
                        x += c - 128
                    else:
                        rgbe[y, x:x + c, ch] = np.frombuffer(data[pos:pos + c], np.uint8); pos += c
                        x += c
        else:
            rgbe[y] = np.frombuffer(data[pos:pos + 4 * w], np.uint8).reshape(w, 4); pos += 4 * w
    return rgbe_to_float(rgbe)
