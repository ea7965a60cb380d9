"""Test-only PIZ *encoder* (the product only reads PIZ): forward wavelet, bitmap / LUT, canonical Huffman
with the 6-bit packed length table -- written from the published OpenEXR PIZ format so that the reader in
psdr_cuda/exr.py can be round-tripped on odd sizes, short last chunks and FLOAT channels, which the
reference's own PIZ file (1024x512 half) does not exercise."""
import heapq
import struct

import numpy as np


def _wenc14(a, b):
    as_ = a.astype(np.int16).astype(np.int32)
    bs = b.astype(np.int16).astype(np.int32)
    ms = (as_ + bs) >> 1
    ds = as_ - bs
    return (ms & 0xFFFF).astype(np.uint16), (ds & 0xFFFF).astype(np.uint16)


def _wenc16(a, b):
    ao = (a.astype(np.int32) + 0x8000) & 0xFFFF
    m = (ao + b.astype(np.int32)) >> 1
    d = ao - b.astype(np.int32)
    m = np.where(d < 0, (m + 0x8000) & 0xFFFF, m)
    return m.astype(np.uint16), (d & 0xFFFF).astype(np.uint16)


def wav2_encode(a, max_value):
    ny, nx = a.shape
    enc = _wenc14 if max_value < (1 << 14) else _wenc16
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        ys = np.arange(0, ny - p2 + 1, p2)
        xs = np.arange(0, nx - p2 + 1, p2)
        if ys.size and xs.size:
            Y, X = np.meshgrid(ys, xs, indexing="ij")
            i00, i01 = enc(a[Y, X], a[Y, X + p])
            i10, i11 = enc(a[Y + p, X], a[Y + p, X + p])
            a[Y, X], a[Y + p, X] = enc(i00, i10)
            a[Y, X + p], a[Y + p, X + p] = enc(i01, i11)
        if nx & p and ys.size:
            x = (xs[-1] + p2) if xs.size else 0
            a[ys, x], a[ys + p, x] = enc(a[ys, x], a[ys + p, x])
        if ny & p:
            y = (ys[-1] + p2) if ys.size else 0
            if xs.size:
                a[y, xs], a[y, xs + p] = enc(a[y, xs], a[y, xs + p])
        p = p2
        p2 <<= 1


class _BitWriter:
    def __init__(self):
        self.acc, self.n, self.out, self.total = 0, 0, bytearray(), 0

    def put(self, nbits, value):
        self.acc = (self.acc << nbits) | value
        self.n += nbits
        self.total += nbits
        while self.n >= 8:
            self.out.append((self.acc >> (self.n - 8)) & 0xFF)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.out.append((self.acc << (8 - self.n)) & 0xFF)
            self.acc, self.n = 0, 0
        return bytes(self.out)


def huf_compress(symbols):
    freq = np.bincount(symbols, minlength=65537).astype(np.int64)
    im = int(np.nonzero(freq)[0][0])
    iM = int(np.nonzero(freq)[0][-1]) + 1                 # run-length pseudo symbol
    freq[iM] = 1
    # Huffman code lengths
    heap = [(int(f), i, (i,)) for i, f in enumerate(freq) if f]
    heapq.heapify(heap)
    length = np.zeros(65537, dtype=np.int64)
    if len(heap) == 1:
        length[heap[0][1]] = 1
    cnt = 70000
    while len(heap) > 1:
        f1, _, s1 = heapq.heappop(heap)
        f2, _, s2 = heapq.heappop(heap)
        for s in s1 + s2:
            length[s] += 1
        heapq.heappush(heap, (f1 + f2, cnt, s1 + s2))
        cnt += 1
    assert length.max() <= 58
    # canonical codes exactly as the decoder derives them
    count = np.bincount(length, minlength=59)
    first = np.zeros(60, dtype=np.int64)
    c = 0
    for l in range(58, 0, -1):
        nc = (c + int(count[l])) >> 1
        first[l] = c
        c = nc
    code = np.zeros(65537, dtype=np.int64)
    nxt = first.copy()
    for s in range(65537):
        l = int(length[s])
        if l:
            code[s] = nxt[l]
            nxt[l] += 1
    # packed table
    tb = _BitWriter()
    i = im
    while i <= iM:
        l = int(length[i])
        if l == 0:
            run = 1
            while i + run <= iM and length[i + run] == 0 and run < 255 + 6:
                run += 1
            if run >= 6:
                tb.put(6, 63); tb.put(8, run - 6)
                i += run
                continue
            if run >= 2:
                tb.put(6, 59 + run - 2)
                i += run
                continue
        tb.put(6, l)
        i += 1
    table = tb.flush()
    db = _BitWriter()
    for s in symbols.tolist():
        db.put(int(length[s]), int(code[s]))
    nbits = db.total
    data = db.flush()
    return struct.pack("<IIIII", im, iM, len(table), nbits, 0) + table + data


def piz_compress_chunk(planes):
    """planes: list of uint16 arrays [nrows, w, size] (size = 1 half, 2 float/uint)"""
    allv = np.concatenate([p.reshape(-1) for p in planes])
    present = np.zeros(65536, dtype=bool)
    present[allv] = True
    present[0] = False
    nz = np.nonzero(present)[0]
    bitmap = np.packbits(present, bitorder="little")
    if nz.size:
        mn, mx = int(nz[0]) >> 3, int(nz[-1]) >> 3
        head = struct.pack("<HH", mn, mx) + bitmap[mn:mx + 1].tobytes()
    else:
        head = struct.pack("<HH", 8191, 0)
    present[0] = True
    fwd = np.cumsum(present) - 1                               # value -> index
    max_value = int(present.sum()) - 1
    coded = []
    for p in planes:
        q = fwd[p].astype(np.uint16)
        for j in range(q.shape[2]):
            comp = np.ascontiguousarray(q[:, :, j])
            wav2_encode(comp, max_value)
            q[:, :, j] = comp
        coded.append(q.reshape(-1))
    huf = huf_compress(np.concatenate(coded).astype(np.int64))
    return head + struct.pack("<i", len(huf)) + huf


def save_exr_piz(path, channels):
    """channels: dict name -> [h, w] array of dtype float16 / float32 / uint32 (written in name order)"""
    names = sorted(channels)
    h, w = channels[names[0]].shape
    ptype = {np.dtype(np.uint32): 0, np.dtype(np.float16): 1, np.dtype(np.float32): 2}

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val
    ch = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", ptype[channels[n].dtype], 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    head = struct.pack("<II", 20000630, 2)
    head += attr("channels", "chlist", ch) + attr("compression", "compression", bytes([4]))
    head += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box)
    head += attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    head += b"\0"
    chunks = []
    for y0 in range(0, h, 32):
        planes = []
        for n in names:
            a = np.ascontiguousarray(channels[n][y0:y0 + 32])
            planes.append(a.view(np.uint16).reshape(a.shape[0], w, a.dtype.itemsize // 2))
        payload = piz_compress_chunk(planes)
        chunks.append(struct.pack("<ii", y0, len(payload)) + payload)
    off = len(head) + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", off)
        off += len(c)
    with open(path, "wb") as f:
        f.write(head + table + b"".join(chunks))
