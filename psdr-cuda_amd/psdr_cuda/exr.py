"""Minimal OpenEXR scan-line reader / writer (replaces tinyexr for BitmapLoader::load_openexr_rgba,
reference src/core/bitmap_loader.cpp:13-53).

Reads single-part scan-line files with NONE / ZIPS / ZIP compression and HALF / FLOAT / UINT
channels (the reference's `test_texture.exr` is 512x512 float ZIP).  PIZ (`ballroom_1k.exr`, the
environment map of SURVEY 8f N2) is not implemented.  Missing A defaults to 1, missing colour
channels to 0, a single luminance channel Y is replicated -- as tinyexr's LoadEXR does.
"""
import struct
import zlib

import numpy as np

_MAGIC = 20000630
_PIXEL = {0: (np.uint32, 4), 1: (np.float16, 2), 2: (np.float32, 4)}
_LINES = {0: 1, 2: 1, 3: 16}


def _unzip(block, expected):
    if len(block) == expected:
        return np.frombuffer(block, dtype=np.uint8)
    raw = np.frombuffer(zlib.decompress(block), dtype=np.uint8)
    # predictor: d[i] = d[i-1] + d[i] - 128 (mod 256)
    d = raw.astype(np.int64)
    d[1:] -= 128
    d = (np.cumsum(d) & 0xFF).astype(np.uint8)
    # de-interleave: first half = even bytes, second half = odd bytes
    n = d.size
    half = (n + 1) // 2
    out = np.empty(n, dtype=np.uint8)
    out[0::2] = d[:half]
    out[1::2] = d[half:]
    return out


def load_exr_rgba(path):
    """Returns (float32 array [h, w, 4] in RGBA order, (w, h))."""
    with open(path, "rb") as f:
        data = f.read()
    magic, version = struct.unpack_from("<II", data, 0)
    if magic != _MAGIC:
        raise RuntimeError("Failed to load EXR (bad magic): " + path)
    if version & 0x1E00:
        raise RuntimeError("Failed to load EXR (tiled / multi-part / deep files are not supported): " + path)
    p = 8
    attrs = {}
    while data[p] != 0:
        e = data.index(b"\0", p); name = data[p:e].decode(); p = e + 1
        e = data.index(b"\0", p); p = e + 1
        size = struct.unpack_from("<i", data, p)[0]; p += 4
        attrs[name] = data[p:p + size]; p += size
    p += 1
    channels = []
    c = attrs["channels"]; q = 0
    while c[q] != 0:
        e = c.index(b"\0", q); cname = c[q:e].decode(); q = e + 1
        ptype = struct.unpack_from("<i", c, q)[0]
        xs, ys = struct.unpack_from("<ii", c, q + 8)
        if xs != 1 or ys != 1:
            raise RuntimeError("Failed to load EXR (sub-sampled channels): " + path)
        channels.append((cname, ptype)); q += 16
    comp = attrs["compression"][0]
    if comp not in _LINES:
        raise RuntimeError("Failed to load EXR (compression %d not supported; NONE/ZIPS/ZIP only): %s" % (comp, path))
    xmin, ymin, xmax, ymax = struct.unpack("<iiii", attrs["dataWindow"])
    w, h = xmax - xmin + 1, ymax - ymin + 1
    lines = _LINES[comp]
    nchunks = (h + lines - 1) // lines
    offsets = struct.unpack_from("<%dQ" % nchunks, data, p)
    row_bytes = sum(_PIXEL[t][1] for _, t in channels) * w
    planes = {n: np.zeros((h, w), dtype=np.float32) for n, _ in channels}
    for off in offsets:
        y, size = struct.unpack_from("<ii", data, off)
        nrows = min(lines, ymax - y + 1)
        raw = _unzip(data[off + 8: off + 8 + size], row_bytes * nrows) if comp else np.frombuffer(data[off + 8: off + 8 + size], dtype=np.uint8)
        pos = 0
        for r in range(nrows):
            for cname, ptype in channels:
                dt, nb = _PIXEL[ptype]
                planes[cname][y - ymin + r] = np.frombuffer(raw[pos:pos + nb * w].tobytes(), dtype=dt).astype(np.float32)
                pos += nb * w
    out = np.zeros((h, w, 4), dtype=np.float32)
    out[..., 3] = 1.0
    if "R" in planes or "G" in planes or "B" in planes:
        for i, n in enumerate("RGB"):
            if n in planes:
                out[..., i] = planes[n]
    elif "Y" in planes:
        out[..., 0] = out[..., 1] = out[..., 2] = planes["Y"]
    elif channels:
        out[..., 0] = out[..., 1] = out[..., 2] = planes[channels[0][0]]
    if "A" in planes:
        out[..., 3] = planes["A"]
    return out, (w, h)


def save_exr_rgb(path, img, compress=True):
    """Writes a float32 RGB scan-line EXR (ZIP or uncompressed).  img: [h, w, 3]."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w = img.shape[:2]

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val
    ch = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for n in "BGR") + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    head = struct.pack("<II", _MAGIC, 2)
    head += attr("channels", "chlist", ch) + attr("compression", "compression", bytes([3 if compress else 0]))
    head += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box)
    head += attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    head += b"\0"
    lines = 16 if compress else 1
    chunks = []
    for y0 in range(0, h, lines):
        rows = img[y0:y0 + lines]
        raw = b"".join(rows[r, :, c].tobytes() for r in range(rows.shape[0]) for c in (2, 1, 0))
        if compress:
            d = np.frombuffer(raw, dtype=np.uint8)
            t = np.concatenate([d[0::2], d[1::2]]).astype(np.int64)
            p = t.copy()
            p[1:] = (t[1:] - t[:-1] + 128 + 256) & 0xFF
            z = zlib.compress(p.astype(np.uint8).tobytes())
            payload = z if len(z) < len(raw) else raw
        else:
            payload = raw
        chunks.append(struct.pack("<ii", y0, len(payload)) + payload)
    table_pos = len(head)
    off = table_pos + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", off); off += len(c)
    with open(path, "wb") as f:
        f.write(head + table + b"".join(chunks))
