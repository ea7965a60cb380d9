"""Minimal OpenEXR scan-line reader / writer (replaces tinyexr for BitmapLoader::load_openexr_rgba,
reference src/core/bitmap_loader.cpp:13-53).

Reads single-part scan-line files with NONE / ZIPS / ZIP / PIZ compression and HALF / FLOAT / UINT
channels (the reference's `test_texture.exr` is 512x512 float ZIP, its environment map
`ballroom_1k.exr` 1024x512 half PIZ).  Missing A defaults to 1, missing colour channels to 0, a single
luminance channel Y is replicated -- as tinyexr's LoadEXR does.

PIZ (OpenEXR's wavelet codec, published format): per 32-line chunk a bitmap of the 16-bit values in use,
a canonical-Huffman coded stream (6-bit packed code lengths with zero runs, an RLE symbol) of the
2-D Haar-like wavelet coefficients of every channel plane, and a look-up table back to pixel values.
"""
import struct
import zlib

import numpy as np

_MAGIC = 20000630
_PIXEL = {0: (np.uint32, 4), 1: (np.float16, 2), 2: (np.float32, 4)}
_LINES = {0: 1, 2: 1, 3: 16, 4: 32}


# ------------------------------------------------------------------------------------ PIZ
class _Bits:
    """MSB-first bit reader over bytes"""

    def __init__(self, data, pos=0):
        self.d, self.p, self.c, self.lc = data, pos, 0, 0

    def get(self, n):
        while self.lc < n:
            self.c = (self.c << 8) | self.d[self.p]
            self.p += 1
            self.lc += 8
        self.lc -= n
        return (self.c >> self.lc) & ((1 << n) - 1)


def _huf_decode(block, n_out):
    """hufUncompress: returns n_out uint16 symbols"""
    im, iM, _table_len, nbits = struct.unpack_from("<IIII", block, 0)
    if im > 65536 or iM > 65536:
        raise RuntimeError("Failed to load EXR (corrupt PIZ Huffman header)")
    # packed code lengths for symbols im..iM (6 bits each, 59..62 = short zero run, 63 = long zero run)
    length = np.zeros(65537, dtype=np.int64)
    br = _Bits(block, 20)
    i = im
    while i <= iM:
        l = br.get(6)
        if l == 63:
            i += br.get(8) + 6
        elif l >= 59:
            i += l - 59 + 2
        else:
            length[i] = l
            i += 1
    pos = br.p                                  # the table ends on a byte boundary
    # canonical codes: for each length, consecutive codes; shorter codes = numerically larger prefixes
    count = np.bincount(length, minlength=59)
    first = np.zeros(60, dtype=np.int64)
    c = 0
    for l in range(58, 0, -1):
        nc = (c + int(count[l])) >> 1
        first[l] = c
        c = nc
    syms = np.nonzero(length)[0]
    order = np.argsort(length[syms], kind="stable")          # symbols in increasing index within a length
    table = {}
    nxt = first.copy()
    for sidx in syms[order]:
        l = int(length[sidx])
        table[(l, int(nxt[l]))] = int(sidx)
        nxt[l] += 1
    # fast table for codes of <= 12 bits
    FAST = 12
    fast_sym = np.full(1 << FAST, -1, dtype=np.int64)
    fast_len = np.zeros(1 << FAST, dtype=np.int64)
    for (l, code), sidx in table.items():
        if l <= FAST:
            lo = code << (FAST - l)
            fast_sym[lo:lo + (1 << (FAST - l))] = sidx
            fast_len[lo:lo + (1 << (FAST - l))] = l
    fast_sym, fast_len = fast_sym.tolist(), fast_len.tolist()
    max_len = int(length.max())
    out = np.empty(n_out, dtype=np.uint16)
    d = block
    end_bits = nbits
    acc, nacc, used, k = 0, 0, 0, 0
    rlc = iM
    while k < n_out:
        while nacc < 32 and pos < len(d):
            acc = (acc << 8) | d[pos]
            pos += 1
            nacc += 8
        if nacc < FAST:
            peek = (acc << (FAST - nacc)) & ((1 << FAST) - 1)
        else:
            peek = (acc >> (nacc - FAST)) & ((1 << FAST) - 1)
        sym, l = fast_sym[peek], fast_len[peek]
        if sym < 0 or l > nacc:
            sym = -1
            for l in range(FAST + 1 if sym < 0 else 1, max_len + 1):
                if l > nacc:
                    break
                code = (acc >> (nacc - l)) & ((1 << l) - 1)
                sym = table.get((l, code), -1)
                if sym >= 0:
                    break
            if sym < 0:
                raise RuntimeError("Failed to load EXR (corrupt PIZ Huffman stream)")
        nacc -= l
        used += l
        if sym == rlc:
            if nacc < 8:
                acc = (acc << 8) | d[pos]
                pos += 1
                nacc += 8
            rep = (acc >> (nacc - 8)) & 0xFF
            nacc -= 8
            used += 8
            if k == 0 or k + rep > n_out:
                raise RuntimeError("Failed to load EXR (corrupt PIZ run)")
            out[k:k + rep] = out[k - 1]
            k += rep
        else:
            out[k] = sym
            k += 1
        acc &= (1 << nacc) - 1
    if used > end_bits:
        raise RuntimeError("Failed to load EXR (PIZ Huffman stream overrun)")
    return out


def _wdec14(l, h):
    ls = l.astype(np.int16).astype(np.int32)
    hs = h.astype(np.int16).astype(np.int32)
    a = ls + (hs & 1) + (hs >> 1)
    return (a & 0xFFFF).astype(np.uint16), ((a - hs) & 0xFFFF).astype(np.uint16)


def _wdec16(l, h):
    m, d = l.astype(np.int32), h.astype(np.int32)
    bb = (m - (d >> 1)) & 0xFFFF
    aa = (d + bb - 0x8000) & 0xFFFF
    return aa.astype(np.uint16), bb.astype(np.uint16)


def _wav2_decode(a, max_value):
    """in-place inverse 2-D wavelet of one plane a[ny, nx] (uint16), wav2Decode"""
    ny, nx = a.shape
    dec = _wdec14 if max_value < (1 << 14) else _wdec16
    n = min(nx, ny)
    p = 1
    while p <= n:
        p <<= 1
    p >>= 1
    p2 = p
    p >>= 1
    while p >= 1:
        ys = np.arange(0, ny - p2 + 1, p2) if ny - p2 >= 0 else np.arange(0)
        xs = np.arange(0, nx - p2 + 1, p2) if nx - p2 >= 0 else np.arange(0)
        if ys.size and xs.size:
            Y, X = np.meshgrid(ys, xs, indexing="ij")
            px, p01, p10, p11 = a[Y, X], a[Y, X + p], a[Y + p, X], a[Y + p, X + p]
            i00, i10 = dec(px, p10)
            i01, i11 = dec(p01, p11)
            a[Y, X], a[Y, X + p] = dec(i00, i01)
            a[Y + p, X], a[Y + p, X + p] = dec(i10, i11)
        if nx & p and ys.size:                       # odd column at x = last block start
            x = (xs[-1] + p2) if xs.size else 0
            a[ys, x], a[ys + p, x] = dec(a[ys, x], a[ys + p, x])
        if ny & p:
            y = (ys[-1] + p2) if ys.size else 0
            if xs.size:
                a[y, xs], a[y, xs + p] = dec(a[y, xs], a[y, xs + p])
        p2 = p
        p >>= 1


def _unpiz(block, channels, w, nrows):
    """channels: list of (name, pixel type); returns the chunk as raw scan-line bytes (like NONE)"""
    sizes = [_PIXEL[t][1] // 2 for _, t in channels]              # uint16 words per pixel
    total = sum(sz * w * nrows for sz in sizes)
    if len(block) == total * 2:
        return np.frombuffer(block, dtype=np.uint8)
    min_nz, max_nz = struct.unpack_from("<HH", block, 0)
    bitmap = np.zeros(8192, dtype=np.uint8)
    pos = 4
    if min_nz <= max_nz:
        bitmap[min_nz:max_nz + 1] = np.frombuffer(block[pos:pos + max_nz - min_nz + 1], dtype=np.uint8)
        pos += max_nz - min_nz + 1
    bits = np.unpackbits(bitmap, bitorder="little")
    bits[0] = 1                                                     # zero is always present
    lut = np.zeros(65536, dtype=np.uint16)
    used = np.nonzero(bits)[0]
    lut[:used.size] = used
    max_value = used.size - 1
    length = struct.unpack_from("<i", block, pos)[0]
    pos += 4
    tmp = _huf_decode(block[pos:pos + length], total)
    planes, o = [], 0
    for sz in sizes:
        n = sz * w * nrows
        pl = tmp[o:o + n].reshape(nrows, w, sz).copy()
        for j in range(sz):
            comp = np.ascontiguousarray(pl[:, :, j])
            _wav2_decode(comp, max_value)
            pl[:, :, j] = comp
        planes.append(lut[pl])
        o += n
    rows = []
    for r in range(nrows):
        for pl in planes:
            rows.append(pl[r].reshape(-1))
    return np.concatenate(rows).astype("<u2").view(np.uint8)



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
        raise RuntimeError("Failed to load EXR (compression %d not supported; NONE/ZIPS/ZIP/PIZ only): %s" % (comp, path))
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
        blk = data[off + 8: off + 8 + size]
        if comp == 4:
            raw = _unpiz(blk, channels, w, nrows)
        else:
            raw = _unzip(blk, row_bytes * nrows) if comp else np.frombuffer(blk, dtype=np.uint8)
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
