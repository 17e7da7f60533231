"""Lossless octree coder for the stride-8 latent coordinates -- pure-Python restatement (test infrastructure).

The reference codes these coordinates by writing an ASCII PLY and calling the external MPEG G-PCC `tmc3` binary
(`model/model.py:388-486`); that binary is absent here, so the build defines its own lossless format:
depth-first octree over Morton-sorted cells, occupancy bits under an LZMA-style adaptive binary range coder with the
context (depth, child position, occupied siblings so far).  This file restates that format for bit-exact checks of the
C++ coder; only round-trip losslessness is a reference-level requirement.
"""
import numpy as np

PROB_BITS, MOVE_BITS, TOP = 11, 5, 1 << 24
M32 = 0xFFFFFFFF


def _morton(x, y, z, depth):
    m = 0
    for b in range(depth):
        m |= (((x >> b) & 1) << (3 * b + 2)) | (((y >> b) & 1) << (3 * b + 1)) | (((z >> b) & 1) << (3 * b))
    return m


def _ctx(level, child, occ):
    return (level * 8 + child) * 8 + min(occ, 7)


class _Enc:
    def __init__(self):
        self.out = bytearray()
        self.low, self.range, self.cache, self.cache_size = 0, M32, 0, 1

    def shift_low(self):
        if (self.low & M32) < 0xFF000000 or (self.low >> 32) != 0:
            temp = self.cache
            while True:
                self.out.append((temp + (self.low >> 32)) & 0xFF)
                temp = 0xFF
                self.cache_size -= 1
                if self.cache_size == 0:
                    break
            self.cache = ((self.low & M32) >> 24) & 0xFF
        self.cache_size += 1
        self.low = ((self.low & M32) << 8) & M32

    def encode(self, probs, i, bit):
        p = probs[i]
        bound = (self.range >> PROB_BITS) * p
        if not bit:
            self.range = bound
            probs[i] = p + (((1 << PROB_BITS) - p) >> MOVE_BITS)
        else:
            self.low += bound
            self.range -= bound
            probs[i] = p - (p >> MOVE_BITS)
        while self.range < TOP:
            self.range = (self.range << 8) & M32
            self.shift_low()

    def flush(self):
        for _ in range(5):
            self.shift_low()


def encode(cells, depth):
    """cells [n,3] unique (x,y,z) in [0, 2^depth) -> bytes (u32 n | u8 depth | range-coder payload)."""
    cells = np.asarray(cells, dtype=np.int64).reshape(-1, 3)
    m = sorted(_morton(int(x), int(y), int(z), depth) for x, y, z in cells)
    out = bytearray(int(len(m)).to_bytes(4, "little")) + bytes([depth])
    if not m:
        return bytes(out)
    probs = [1 << (PROB_BITS - 1)] * (depth * 64)
    rc = _Enc()

    def node(lo, hi, level):
        if level == depth:
            return
        shift = 3 * (depth - 1 - level)
        start, p = [], lo
        for c in range(8):
            start.append(p)
            while p < hi and ((m[p] >> shift) & 7) == c:
                p += 1
        start.append(hi)
        occ = 0
        for c in range(8):
            bit = 1 if start[c + 1] > start[c] else 0
            rc.encode(probs, _ctx(level, c, occ), bit)
            occ += bit
        for c in range(8):
            if start[c + 1] > start[c]:
                node(start[c], start[c + 1], level + 1)

    node(0, len(m), 0)
    rc.flush()
    return bytes(out + rc.out)


def decode(data):
    """bytes -> ([n,3] int32 cells in Morton order, depth)."""
    n = int.from_bytes(data[:4], "little")
    depth = data[4]
    if n == 0:
        return np.zeros((0, 3), np.int32), depth
    buf, pos = data, 5
    rng, code = M32, 0

    def nxt():
        nonlocal pos
        b = buf[pos] if pos < len(buf) else 0
        pos += 1
        return b
    for _ in range(5):
        code = ((code << 8) | nxt()) & M32
    probs = [1 << (PROB_BITS - 1)] * (depth * 64)
    out = []

    def dec(i):
        nonlocal rng, code
        p = probs[i]
        bound = (rng >> PROB_BITS) * p
        if code < bound:
            rng = bound
            probs[i] = p + (((1 << PROB_BITS) - p) >> MOVE_BITS)
            bit = 0
        else:
            code -= bound
            rng -= bound
            probs[i] = p - (p >> MOVE_BITS)
            bit = 1
        while rng < TOP:
            rng = (rng << 8) & M32
            code = ((code << 8) | nxt()) & M32
        return bit

    def node(prefix, level):
        if level == depth:
            out.append(prefix)
            return
        bits, occ = [], 0
        for c in range(8):
            b = dec(_ctx(level, c, occ))
            bits.append(b)
            occ += b
        assert occ > 0, "corrupt stream"
        for c in range(8):
            if bits[c]:
                node((prefix << 3) | c, level + 1)

    node(0, 0)
    assert len(out) == n
    cells = np.zeros((n, 3), np.int32)
    for i, mm in enumerate(out):
        for b in range(depth):
            cells[i, 0] |= ((mm >> (3 * b + 2)) & 1) << b
            cells[i, 1] |= ((mm >> (3 * b + 1)) & 1) << b
            cells[i, 2] |= ((mm >> (3 * b)) & 1) << b
    return cells, depth
