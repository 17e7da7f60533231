"""Bitstream container + lossless latent-coordinate coding.

Field order follows the reference's `UnifiedModel.save_bitstream / load_bitstream` (`model/model.py:253-385`):
    "PCCB" | uint16 version (2) | uint16 flags (0)                                   -- format word (round 3)
    int32 num_blocks
    per block:  int32 Nz | int32 len(points stream) | float64 q_g | float64 q_a | int32 len(y string) |
                int32 len(z string) | int32 k1 | int32 k2 | int32 k3 |
                uint32 stream geometry = channel groups of the y string << 16 | of the z string (0: not recorded) |
                points stream | y string | z string
The stream geometry makes a file independent of the coder's tuning constants (`EntropyModel.STREAM_SYMBOLS` decides how many
channels share a GPU stream; a decoder built with another value would otherwise cut the container differently): the strings
come back as `StreamBytes` carrying it.  Version-1 files (no format word, no geometry) are still read.
Integers / floats are little-endian here (the reference writes them through the `bitstream` package, which is not
available to check its byte order).  The points stream is the octree coder of libpcc_hip (`pcc_octree_*_host`) instead
of an ASCII-PLY round trip through the external G-PCC `tmc3` binary (`model/model.py:388-486`); a second header
(origin, pitch, depth: 17 bytes) in front of the octree payload carries what `tmc3` keeps in its own syntax.
"""
import ctypes as C
import struct

import numpy as np
import torch

from . import lib as L


MAGIC, VERSION = b"PCCB", 2


class StreamBytes(bytes):
    """A coded string that remembers its stream geometry (channel groups of the `pcc_streams` container); otherwise bytes."""

    groups = 0

    def __new__(cls, data, groups=0):
        b = super().__new__(cls, data)
        b.groups = int(groups)
        return b


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def encode_points(coords, pitch=8):
    """coords: [n,4] (b,x,y,z) or [n,3] integer tensor/array of lattice points (multiples of `pitch`) -> bytes."""
    c = coords.detach().cpu().numpy() if torch.is_tensor(coords) else np.asarray(coords)
    xyz = np.ascontiguousarray(c[:, -3:]).astype(np.int64)
    n = xyz.shape[0]
    if n == 0:
        return struct.pack("<iiiiB", 0, 0, 0, pitch, 1) + struct.pack("<IB", 0, 1)
    origin = xyz.min(axis=0)
    cells = (xyz - origin) // pitch
    if np.any((xyz - origin) % pitch):
        raise L.PccError(f"latent coordinates are not on a pitch-{pitch} lattice")
    depth = max(int(np.ceil(np.log2(int(cells.max()) + 1))), 1)
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    lib = L.load()
    cap = lib.pcc_octree_max_bytes(n, depth)
    out = np.zeros(cap, np.uint8)
    nb = C.c_int64(0)
    L.check(lib.pcc_octree_encode_host(_np_ptr(cells), n, depth, _np_ptr(out), cap, C.byref(nb)), "pcc_octree_encode_host")
    return struct.pack("<iiiiB", int(origin[0]), int(origin[1]), int(origin[2]), pitch, depth) + out[:nb.value].tobytes()


def decode_points(data):
    """bytes -> int32 [n,3] array of (x,y,z), ascending (x,y,z) (= canonical order of a one-batch set)."""
    if len(data) < 17 + 5:
        raise L.PccError("latent-coordinate stream truncated")
    ox, oy, oz, pitch, _depth = struct.unpack_from("<iiiiB", data, 0)
    if pitch <= 0 or pitch > (1 << 15):
        raise L.PccError(f"latent-coordinate stream: implausible pitch {pitch}")
    body = np.frombuffer(data, np.uint8, offset=17).copy()
    lib = L.load()
    n, depth = C.c_int64(0), C.c_int32(0)
    L.check(lib.pcc_octree_decode_host(_np_ptr(body), len(body), None, 0, C.byref(n), C.byref(depth)), "pcc_octree_decode_host")
    # the point count comes from the stream header: bound it before allocating for it.  The adaptive range coder's cost per
    # occupancy decision saturates near 0.02 bit, so a fully dense block codes at well over 100 points per payload byte: the
    # bound is what the lattice can hold (8^depth cells) and a loose 4096 points per byte, not a rate assumption
    if n.value < 0 or n.value > min(8 ** min(int(depth.value), 20), 4096 * max(len(body), 1) + 4096):
        raise L.PccError(f"latent-coordinate stream: header claims {n.value} points for {len(body)} payload bytes at depth {depth.value}")
    cells = np.zeros((max(n.value, 1), 3), np.int32)
    L.check(lib.pcc_octree_decode_host(_np_ptr(body), len(body), _np_ptr(cells), n.value, C.byref(n), C.byref(depth)),
            "pcc_octree_decode_host")
    xyz = cells[:n.value].astype(np.int64) * pitch + np.array([ox, oy, oz])
    order = np.lexsort((xyz[:, 2], xyz[:, 1], xyz[:, 0]))
    return xyz[order].astype(np.int32)


def save_bitstream(path, blocks_coordinates, blocks_strings, blocks_shapes, blocks_k, blocks_q):
    """`UnifiedModel.save_bitstream` (`model/model.py:253-311`)."""
    out = bytearray(MAGIC + struct.pack("<HHi", VERSION, 0, len(blocks_coordinates)))
    for coords, strings, shape, k, q in zip(blocks_coordinates, blocks_strings, blocks_shapes, blocks_k, blocks_q):
        pts = encode_points(coords)
        qv = q.detach().cpu().double().reshape(-1)
        out += struct.pack("<iidd", int(shape[0]), len(pts), float(qv[0]), float(qv[1]))
        for s in strings:
            out += struct.pack("<i", len(s[0]))
        for ks in k:
            out += struct.pack("<i", int(ks[0]))
        gy, gz = (min(int(getattr(s[0], "groups", 0)), 0xFFFF) for s in strings)
        out += struct.pack("<I", (gy << 16) | gz)
        out += pts
        for s in strings:
            out += s[0]
    with open(path, "wb") as f:
        f.write(bytes(out))
    return len(out)


def load_bitstream(path):
    """`UnifiedModel.load_bitstream` (`model/model.py:314-385`): coordinates come back as [n,3] int32 (x,y,z)."""
    data = open(path, "rb").read()
    if len(data) < 4:
        raise L.PccError("bitstream file truncated")
    version, off = 1, 0
    if data[:4] == MAGIC:
        if len(data) < 12:
            raise L.PccError("bitstream file truncated")
        version, _flags = struct.unpack_from("<HH", data, 4)
        if version != VERSION:
            raise L.PccError(f"bitstream file: format version {version}, this build reads {VERSION} (and headerless version-1 files)")
        off = 8
    (nblocks,), off = struct.unpack_from("<i", data, off), off + 4
    if nblocks < 0 or nblocks * 44 > len(data):
        raise L.PccError(f"bitstream file: implausible block count {nblocks}")
    coords, strings, shapes, ks, qs = [], [], [], [], []
    for _ in range(nblocks):
        if off + struct.calcsize("<iiddiiiii") + (4 if version >= 2 else 0) > len(data):
            raise L.PccError("bitstream file truncated inside a block header")
        nz, lp, qg, qa, ly, lz, k1, k2, k3 = struct.unpack_from("<iiddiiiii", data, off)
        off += struct.calcsize("<iiddiiiii")
        geom = 0
        if version >= 2:
            (geom,), off = struct.unpack_from("<I", data, off), off + 4
        if min(nz, lp, ly, lz, k1, k2, k3) < 0 or off + lp + ly + lz > len(data):
            raise L.PccError("bitstream file: block header with negative or oversized lengths")
        pts = data[off:off + lp]; off += lp
        ys = StreamBytes(data[off:off + ly], geom >> 16); off += ly
        zs = StreamBytes(data[off:off + lz], geom & 0xFFFF); off += lz
        coords.append(torch.from_numpy(decode_points(pts)))
        strings.append([[ys], [zs]])
        shapes.append([nz])
        ks.append([[k1], [k2], [k3]])
        qs.append(torch.tensor([[qg, qa]], dtype=torch.float32))
    if off != len(data):
        raise L.PccError("trailing bytes in bitstream file")
    return coords, strings, shapes, ks, qs
