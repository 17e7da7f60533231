"""Coordinate sets, kernel maps and the functional operator layer over libpcc_hip.

Everything here runs on the GPU through the C ABI (include/pcc_hip.h).  A `CoordSet` is a canonical
(strictly ascending) packed-key array plus host-side bookkeeping; derived sets (stride, generative
expansion) and kernel maps are cached on the set they derive from, which replaces MinkowskiEngine's
coordinate manager.
"""
import ctypes as C

import itertools
import os
import weakref

import torch

from . import lib as L

BIAS = 1 << 15


def _field_mask(lo, hi):
    a, b = lo + BIAS, hi + BIAS
    return (1 << (a ^ b).bit_length()) - 1


class Bounds:
    """Conservative host-side coordinate bounds of a set; they select which radix digits can differ."""

    def __init__(self, bmax, lo, hi):
        self.bmax, self.lo, self.hi = int(bmax), tuple(int(v) for v in lo), tuple(int(v) for v in hi)
        if min(self.lo) < -BIAS + 64 or max(self.hi) >= BIAS - 64 or self.bmax >= (1 << 15):
            raise L.PccError(f"coordinates out of the 16-bit key range: lo={self.lo} hi={self.hi} batch={self.bmax}")

    def bit_mask(self):
        m = ((1 << self.bmax.bit_length()) - 1) << 48
        for i, sh in enumerate((32, 16, 0)):
            m |= _field_mask(self.lo[i], self.hi[i]) << sh
        return m & 0xFFFFFFFFFFFFFFFF

    def strided(self, m):
        return Bounds(self.bmax, [(v // m) * m for v in self.lo], [(v // m) * m for v in self.hi])

    def expanded(self, ksize, step):
        lo_off = -((ksize - 1) // 2) if ksize % 2 else 0
        hi_off = (ksize - 1) // 2 if ksize % 2 else ksize - 1
        return Bounds(self.bmax, [v + lo_off * step for v in self.lo], [v + hi_off * step for v in self.hi])


class KernelMap:
    """Device-resident kernel map: header, neighbour table, optional position->row list."""

    def pairs(self):
        """Number of (in,out) pairs P (FLOP = 2*P*Cin*Cout); reads one device int64."""
        if self._pairs is None:
            self._pairs = int(self.d_pairs.item()) if self.d_pairs is not None else -1
        return self._pairs

    def pair_plan_begin(self):
        """Queue the ranking pass of the pair plan; returns a Pending (see `resolve`) that completes it."""
        if hasattr(self, "_plan"):
            return _ready(self._plan)
        if hasattr(self, "_plan_pending"):
            return self._plan_pending
        if not (self.rows is None and not self.transposed and self.n_out > 0 and self.K * self.n_out + self.K * 128 < (1 << 31)):
            self._plan = None
            return _ready(None)
        dev = self.hdr.device
        lib = L.load()
        pos = torch.empty(self.K * self.n_out, dtype=torch.int32, device=dev)
        pstart = torch.empty(self.K + 1, dtype=torch.int32, device=dev)
        info = L.counter(3)
        ws = L.workspace(lib.pcc_pair_plan_ws_bytes(self.n_out, self.K), dev)
        L.call("pcc_pair_plan_rank", L.ptr(self.nbr), self.n_out, self.K, L.ptr(pos), L.ptr(pstart), L.ptr(info),
               L.ptr(ws), ws.numel(), L.stream())

        def finish(v):
            padded, _, pairs = (int(x) for x in v)
            if self._pairs is None:
                self._pairs = pairs
            self._plan = None
            if pairs <= PAIR_MAX_DENSITY * self.K * self.n_out:
                pair_in = torch.empty(max(padded, 1), dtype=torch.int32, device=dev)
                tile_k = torch.empty(max(padded // 128, 1), dtype=torch.int32, device=dev)
                L.call("pcc_pair_plan_fill", L.ptr(self.nbr), L.ptr(pos), L.ptr(pstart), self.n_out, self.K, padded,
                       L.ptr(pair_in), L.ptr(tile_k), L.stream())
                self._plan = (pos, pair_in, tile_k, info, padded)
            del self._plan_pending
            return self._plan
        self._plan_pending = Pending(info, finish)
        return self._plan_pending

    def pair_plan(self):
        """Compacted pair lists of a conv map (pcc_conv_fwd_pairs), built on first use; None when the map is too dense
        for the pair form to pay (or is not a plain conv map)."""
        if not hasattr(self, "_plan"):
            resolve(self.pair_plan_begin())
        return self._plan

    def dense(self):
        """[K, n_out] int32 table with -1 holes (tests)."""
        out = torch.empty((self.K, self.n_out), dtype=torch.int32, device=self.hdr.device)
        if self.n_out:
            L.call("pcc_map_to_dense", L.ptr(self.hdr), L.ptr(self.nbr), L.ptr(self.rows), self.n_out, self.K,
                   L.ptr(out), L.stream())
        return out


PAIR_MIN_K = 64          # pair-list convolution: kernels with at least this many offsets ...
PAIR_MAX_DENSITY = 0.5   # ... whose map holds at most this fraction of the K * n_out slots ...
PAIR_MIN_CIN = 64        # ... and enough input channels to pay for writing and re-reading T (cin/4 FLOP per byte)
COUNT_PAIRS = False   # bench.py switches this on for its FLOP accounting pass
MORTON_MIN_ROWS = 1 << 62    # Z-curve visiting order of conv maps: measured no gain on MI355X (round 1), off by default
USE_CSR = True        # generative expansion also emits the transposed map as CSR pair lists (False: class map + lookup)
CANON_BY_GRID = True   # canonical order of user rows through the bitmap (False: radix sort + unique)
EXPAND_BY_GRID = True  # generative expansion through the bitmaps (False: 32-bit cell radix sort with pair ids as payload)
STRIDE_BY_GRID = True  # strided sets read out of the coarse occupancy bitmap (False: mask + radix sort + unique)
USE_GRID = True       # neighbour lookup through the bitmap+rank grid index (False: binary search; tests run both)
GRID_MAX_BYTES = 8 << 30
BAND_TILES = os.environ.get("PCC_BAND_TILES", "1") != "0"      # stencil kernels over large sets visit their tiles band by band (L2 locality of the dx = +-1 slabs)
BAND_MIN_ROWS = 1 << 20
BAND_COUNT = 16
T_Z_FASTEST = os.environ.get("PCC_T_Z_FASTEST", "1") != "0"        # composite levels: products laid out [row][kx][ky][kz][c]
T_CHUNKED = os.environ.get("PCC_T_CHUNKED", "0") != "0"            # composite levels: per-pair products staged in cache-sized chunks
#   (measured round 2: bit-identical, 10 GB less memory, but +3.5 ms per step -- 125 chunk pairs of launches, children near
#   chunk borders visited twice, and the Infinity Cache does not speed the gather up enough to pay for it: off)
T_CHUNKED_MIN_BYTES = 256 << 20
BATCH_BOUNDS = os.environ.get("PCC_BATCH_BOUNDS", "1") != "0"   # batched sets: per-batch row ranges of an expanded set read with its size
CSR_SLOTS = os.environ.get("PCC_CSR_SLOTS", "1") != "0"      # composite levels: 7-wide pair lists in one pass (per-workgroup slots)
STENCIL_FROM_GRID = os.environ.get("PCC_STENCIL_FROM_GRID", "1") != "0"   # composite levels: 3x3x3 neighbours from the bitmap, no nbr table
HEAD_FUSED = os.environ.get("PCC_HEAD_FUSED", "1") != "0"      # occupancy heads with <= 16 hidden channels: conv + ReLU + projection in one kernel


_SET_SERIAL = itertools.count()


class Pending:
    """A size the device is still computing (rows of a derived coordinate set, pairs of a list, ...).  The launches are
    queued; `finish(values)` runs once the counter has been read.  `resolve` reads the counters of ANY number of pending
    sizes with one device->host copy: the host reads of a step are batched per level instead of one per derived object
    (round 2: ~33 reads per step, each leaving the GPU idle until the host had queued the next kernels)."""

    __slots__ = ("counter", "finish", "result", "done", "keys_out", "cap")

    def __init__(self, counter, finish):
        self.counter, self.finish, self.result, self.done = counter, finish, None, False


def resolve(*pendings):
    """Read every pending size in one go and complete the objects; returns their results in order.  A Pending's counter
    may be a list of counters (its finish then receives the list of their values)."""
    todo = [p for p in pendings if p is not None and not p.done]
    if todo:
        flat = []
        for p in todo:
            flat.extend(p.counter if isinstance(p.counter, list) else [p.counter])
        vals = L.read_many(flat)
        at = 0
        for p in todo:
            if isinstance(p.counter, list):
                v = vals[at:at + len(p.counter)]
                at += len(p.counter)
            else:
                v = vals[at]
                at += 1
            p.result = p.finish(v)
            p.done = True
    return [p.result if p is not None else None for p in pendings]


def _ready(value):
    p = Pending(None, None)
    p.result, p.done = value, True
    return p


class CoordSet:
    """Canonical coordinate set at one tensor stride."""

    def __init__(self, keys, n, tensor_stride, bounds):
        self.keys = keys                    # int64 [>=n] device, ascending
        self.n = int(n)
        self.ts = int(tensor_stride)
        self.bounds = bounds
        self._C = None
        self._derived = {}
        self._maps = {}
        self._grid = None
        self._bands = None
        self._on_lattice = False            # every row a multiple of the tensor stride: sets derived by the library are, user sets are checked
        self.uid = next(_SET_SERIAL)

    @property
    def device(self):
        return self.keys.device

    def coords(self):
        """int32 [n,4] (b,x,y,z) view, tagged so SparseTensor(coordinates=x.C) re-wraps in O(1).  The tensor refers to
        this set strongly, the set remembers the tensor only weakly: a strong reference both ways is a cycle that only
        the cyclic collector frees, and a step's maps (GBs) would then outlive the step by several steps."""
        c = self._C() if self._C is not None else None
        if c is None:
            c = torch.empty((self.n, 4), dtype=torch.int32, device=self.device)
            if self.n:
                L.call("pcc_keys_unpack", L.ptr(self.keys), self.n, L.ptr(c), L.stream())
            c._pcc_cset = self
            c._pcc_perm = None
            self._C = weakref.ref(c)
        return c

    def grid(self):
        """Occupancy bitmap + rank over the bounding lattice (the lookup structure of every map whose input is this
        set).  Returns (bits, rank, h_grid) or None when the lattice would be unreasonably large."""
        if self._grid is None:
            b = self.bounds
            dims = [(b.hi[i] - b.lo[i]) // self.ts + 1 for i in range(3)]
            h = (C.c_int32 * 8)(b.lo[0], b.lo[1], b.lo[2], dims[0], dims[1], dims[2], self.ts, b.bmax + 1)
            words = L.load().pcc_grid_words(h)
            if self.n == 0 or words >= (1 << 31) or words * 12 > GRID_MAX_BYTES:
                self._grid = False
            else:
                bits = torch.empty(words, dtype=torch.int64, device=self.device)
                rank = torch.empty(words, dtype=torch.int32, device=self.device)
                ws = L.workspace(L.load().pcc_grid_ws_bytes(words), self.device)
                L.call("pcc_grid_build", L.ptr(self.keys), self.n, h, L.ptr(bits), L.ptr(rank), L.ptr(ws), ws.numel(),
                       L.stream())
                self._grid = (bits, rank, h)
        return self._grid or None

    def band_tiles(self):
        """Band-ordered tile table (tiles int32, n_tiles int32[1]) for the stencil kernels over this set, or None when
        the set is small (its neighbour slabs fit L2 anyway), batched, or too large for the 27-bit row field."""
        if self._bands is None:
            self._bands = False
            b = self.bounds
            if BAND_TILES and b.bmax == 0 and BAND_MIN_ROWS <= self.n < (1 << 27):
                lib = L.load()
                nx = (b.hi[0] - b.lo[0]) // self.ts + 1
                ny = (b.hi[1] - b.lo[1]) // self.ts + 1
                nb = max(1, min(BAND_COUNT, ny))
                cap = lib.pcc_band_tiles_cap(self.n, nx, nb)
                tiles = torch.empty(cap, dtype=torch.int32, device=self.device)
                n_tiles = torch.zeros(1, dtype=torch.int32, device=self.device)
                ws = L.workspace(lib.pcc_band_tiles_ws_bytes(nx, nb), self.device)
                L.call("pcc_band_tiles_build", L.ptr(self.keys), self.n, b.lo[0], nx, b.lo[1], ny, self.ts, nb, L.ptr(tiles),
                       cap, L.ptr(n_tiles), L.ptr(ws), ws.numel(), L.stream())
                self._bands = (tiles, n_tiles)
        return self._bands or None

    # ---- derived sets ------------------------------------------------------------------------
    def stride_begin(self, new_stride, known_n=None, keys=None, n_keys=None, d_n=None):
        """Queue the strided set unique(floor(c/m)*m) (a2-i); returns a Pending that yields the CoordSet.
        known_n: the row count when the caller has it (the decoder knows the hyper-latent's from the bitstream): the set
        is complete without a read.  keys / n_keys: take the rows from this key array instead (ANY order, duplicates
        allowed -- marking a coarse cell is idempotent), e.g. the un-canonicalised user rows or a finer ancestor; d_n: a
        counter() holding the number of valid rows of `keys` when the host has not read it yet (n_keys = capacity)."""
        key = ("stride", new_stride)
        if key in self._derived:
            return _ready(self._derived[key])
        dev = self.device
        lib = L.load()
        src, n_src = (self.keys, self.n) if keys is None else (keys, n_keys)
        nb = self.bounds.strided(new_stride)
        dims = [(nb.hi[i] - nb.lo[i]) // new_stride + 1 for i in range(3)]
        h = (C.c_int32 * 8)(nb.lo[0], nb.lo[1], nb.lo[2], dims[0], dims[1], dims[2], new_stride, nb.bmax + 1)
        words = lib.pcc_grid_words(h)
        cnt = L.counter()
        cap = max(min(self.n, n_src), 1)
        out = torch.empty(cap, dtype=torch.int64, device=dev)
        by_grid = USE_GRID and STRIDE_BY_GRID and n_src > 0 and words < (1 << 31) and words * 12 <= GRID_MAX_BYTES
        if by_grid:
            # through the occupancy bitmap of the coarse lattice: no sort, and the coarse set's grid index for free
            bits = torch.empty(words, dtype=torch.int64, device=dev)
            rank = torch.empty(words, dtype=torch.int32, device=dev)
            ws = L.workspace(lib.pcc_grid_ws_bytes(words), dev)
            L.call("pcc_coords_stride_grid", L.ptr(src), n_src, L.cptr(d_n) if d_n is not None else None, h, L.ptr(bits), L.ptr(rank), L.ptr(out),
                   L.cptr(cnt), L.ptr(ws), ws.numel(), L.stream())
        else:
            if keys is not None:
                raise L.PccError("stride_begin(keys=...) needs the bitmap path")
            ws = L.workspace(lib.pcc_stride_ws_bytes(self.n), dev)
            L.call("pcc_coords_stride", L.ptr(self.keys), self.n, new_stride, self.bounds.bit_mask(), L.ptr(out),
                   L.cptr(cnt), L.ptr(ws), ws.numel(), L.stream())

        def finish(v):
            n = int(v[0])
            cs = CoordSet(out, n, new_stride, nb)
            cs._on_lattice = True
            if by_grid:
                cs._grid = (bits, rank, h)
            self._derived[key] = cs
            return cs
        if known_n is not None:
            return _ready(finish([known_n]))
        p = Pending(cnt, finish)
        p.keys_out, p.cap = out, cap
        return p

    def stride(self, new_stride):
        """Output set of a strided conv (a2-i): unique(floor(c/m)*m)."""
        key = ("stride", new_stride)
        if key not in self._derived:
            resolve(self.stride_begin(new_stride))
        return self._derived[key]

    def stride_chain_begin(self, strides, keys=None, n_keys=None):
        """Queue the whole chain self -> stride(s0) -> stride(s1) -> ... (the analysis transform's and the hyper-analysis'
        output sets) without a host read between the links: each link takes its rows from the previous link's output
        buffer, whose row count stays on the device (`pcc_coords_stride_grid` d_n).  One Pending; it links the chain into
        the `_derived` caches the convolution modules look the sets up in."""
        if not (USE_GRID and STRIDE_BY_GRID and self.n > 0):
            return None
        src, n_src = (self.keys, self.n) if keys is None else (keys, n_keys)
        parts = []
        d_n = None
        for m in strides:
            t = CoordSet(self.keys, self.n, self.ts, self.bounds)          # carrier of the bounds; its own cache is discarded
            pend = t.stride_begin(m, keys=src, n_keys=n_src, d_n=d_n)
            parts.append((m, pend))
            # the next link reads the rows this one writes, with their count still on the device (capacity = this link's)
            src, n_src, d_n = pend.keys_out, pend.cap, pend.counter

        def finish(vals):
            parent = self
            for (m, pend), v in zip(parts, vals):
                if not pend.done:
                    pend.result, pend.done = pend.finish(v), True
                parent._derived[("stride", m)] = pend.result
                parent = pend.result
            return parent
        return Pending([p.counter for _, p in parts], finish)

    def csr_for(self, out_keys, n_out, ksize, ts_out, zk=False, total=None, slots=False):
        """CSR pair lists (first[n_out+1], pair_ids) of the transposed conv from this set onto the GIVEN output rows
        (any canonical subset of the lattice at pitch ts_out), built by probing this set's grid index.
        total: a counter() that receives the number of pairs (to be read together with other sizes, `resolve`).
        slots: the one-pass form (`pcc_coords_expand_grid_csr_slots`): returns (first, pair_ids, wg_end) -- row o's list is
        pair_ids[first[o] : first[o + 1]], or up to wg_end[o // 256] for the last row of every 256; consumed by
        `convt_forward_csr_grid`."""
        g = self.grid()
        if not g:
            raise L.PccError("csr_for needs the grid index of the input set")
        dev = self.device
        K = ksize ** 3
        if n_out == 0:
            return torch.zeros(1, dtype=torch.int32, device=dev), torch.empty(1, dtype=torch.int32, device=dev)
        if slots:
            elems = L.load().pcc_expand_grid_csr_slot_elems(n_out, ksize)
            if ksize in (5, 7) and self.ts >= 2 * ts_out and 0 < elems < (1 << 31):
                first = torch.empty(n_out, dtype=torch.int32, device=dev)
                pair_ids = torch.empty(elems, dtype=torch.int32, device=dev)         # worst-case capacity; only the written part is touched
                wg_end = torch.empty((n_out + 255) // 256, dtype=torch.int32, device=dev)
                L.call("pcc_coords_expand_grid_csr_slots", L.ptr(out_keys), n_out, ksize, ts_out, L.ptr(g[0]), L.ptr(g[1]), g[2], self.n,
                       L.ptr(first), L.ptr(pair_ids), L.ptr(wg_end), L.cptr(total) if total is not None else None, 1 if zk else 0,
                       L.stream())
                return first, pair_ids, wg_end
        first = torch.empty(n_out + 1, dtype=torch.int32, device=dev)
        pair_ids = torch.empty(max(self.n * K, 1), dtype=torch.int32, device=dev)
        ws = L.workspace(L.load().pcc_expand_grid_csr_ws_bytes(n_out), dev)
        # zk: pair ids number the kernel offsets z fastest (for product buffers laid out [row][kx][ky][kz][c])
        L.call("pcc_coords_expand_grid_csr_zk" if zk else "pcc_coords_expand_grid_csr", L.ptr(out_keys), n_out, ksize, ts_out,
               L.ptr(g[0]), L.ptr(g[1]), g[2], self.n, L.ptr(first), L.ptr(pair_ids), L.cptr(total) if total is not None else None,
               L.ptr(ws), ws.numel(), L.stream())
        return first, pair_ids

    def _expand_lattice(self, ksize, ts_out):
        ob = self.bounds.expanded(ksize, ts_out)
        dims = [(ob.hi[i] - ob.lo[i]) // ts_out + 1 for i in range(3)]
        cells = (ob.bmax + 1) * dims[0] * dims[1] * dims[2]
        h = (C.c_int32 * 8)(ob.lo[0], ob.lo[1], ob.lo[2], dims[0], dims[1], dims[2], ts_out, ob.bmax + 1)
        return ob, cells, h, L.load().pcc_grid_words(h)

    def _expand_by_grid(self, ksize, ts_out):
        K = ksize ** 3
        _, _, _, words = self._expand_lattice(ksize, ts_out)
        if (USE_CSR and USE_GRID and EXPAND_BY_GRID and self.n > 0 and ksize in (2, 3, 5) and words < (1 << 31)
                and words * 12 <= GRID_MAX_BYTES and self.n * K < (1 << 31)):
            return self.grid()
        return None

    def expand_begin(self, ksize, ts_out, want_csr=True):
        """Queue the generative expansion through the bitmaps; returns a Pending yielding the CoordSet, or None when this
        set does not take the bitmap path (then `expand` does everything, with its own reads).  An even kernel of width 2
        on a pitch-2*ts_out set makes exactly 8 distinct children per row: that count needs no read."""
        key = ("expand", ksize, ts_out)
        if key in self._derived:
            return _ready(self.expand(ksize, ts_out, want_csr))
        if not self._expand_by_grid(ksize, ts_out):
            return None
        dev = self.device
        lib = L.load()
        ob, cells, h, words = self._expand_lattice(ksize, ts_out)
        m = self.n * ksize ** 3
        # mark the output bitmap, read the set back out, then build the pair lists by probing this set's grid
        bits = torch.empty(words, dtype=torch.int64, device=dev)
        rank = torch.empty(words, dtype=torch.int32, device=dev)
        out = torch.empty(min(m, cells), dtype=torch.int64, device=dev)
        cnt = L.counter()
        ws = L.workspace(lib.pcc_grid_ws_bytes(words), dev)
        L.call("pcc_coords_expand_grid", L.ptr(self.keys), self.n, ksize, h, L.ptr(bits), L.ptr(rank), L.ptr(out),
               L.cptr(cnt), L.ptr(ws), ws.numel(), L.stream())
        # batched sets: the per-batch row ranges of the new set (top-k runs per batch) come back with its size
        entries = ob.bmax + 2 if (BATCH_BOUNDS and 0 < ob.bmax <= 10) else 0
        if entries:
            seg_a, seg_b, seg_c = L.counter(4), L.counter(4), L.counter(4)
            L.call("pcc_batch_bounds", L.ptr(out), L.cptr(cnt), 0, entries, L.cptr(seg_a), L.cptr(seg_b), L.cptr(seg_c), L.stream())

        def finish(v):
            segs = None
            if v and isinstance(v[0], (list, tuple)):            # [[n], ranges 0-3, ranges 4-7, ranges 8-11]
                segs = (list(v[1]) + list(v[2]) + list(v[3]))[:entries]
                v = v[0]
            n = int(v[0])
            cs = CoordSet(out[:n].clone() if n < out.numel() // 2 else out, n, ts_out, ob)
            cs._grid = (bits, rank, h)
            cs._on_lattice = self._on_lattice
            if segs is not None:
                cs._derived["segments"] = [int(x) for x in segs]
            self._derived[key] = cs
            if want_csr:
                self._derived[("csr", ksize, ts_out)] = self.csr_for(cs.keys, n, ksize, ts_out)
            return cs
        if ksize == 2 and self.ts == 2 * ts_out and self._on_lattice:
            return _ready(finish([8 * self.n]))
        return Pending([cnt, seg_a, seg_b, seg_c] if entries else cnt, finish)

    def expand(self, ksize, ts_out, want_csr=True):
        """Output set of a generative transposed conv (a3-i): unique{c + off_k*ts_out}, plus (want_csr) the transposed
        map in CSR form (`csr_map`)."""
        key = ("expand", ksize, ts_out)
        if key in self._derived and want_csr and USE_CSR and ("csr", ksize, ts_out) not in self._derived \
                and self._derived[key]._grid and self.grid():
            cs = self._derived[key]                       # the set was built without its pair lists: add them
            self._derived[("csr", ksize, ts_out)] = self.csr_for(cs.keys, cs.n, ksize, ts_out)
        if key not in self._derived:
            pend = self.expand_begin(ksize, ts_out, want_csr)
            if pend is not None:
                return resolve(pend)[0]
            dev = self.device
            K = ksize ** 3
            ob, cells, h, words = self._expand_lattice(ksize, ts_out)
            if USE_CSR and self.n > 0 and cells <= 0xFFFFFFFF and self.n * K < (1 << 31):
                m = self.n * K
                out = torch.empty(m, dtype=torch.int64, device=dev)
                pair_ids = torch.empty(m, dtype=torch.int32, device=dev)
                first = torch.empty(m + 1, dtype=torch.int32, device=dev)
                cnt = L.counter()
                ws = L.workspace(L.load().pcc_expand_csr_ws_bytes(self.n, ksize), dev)
                L.call("pcc_coords_expand_csr", L.ptr(self.keys), self.n, ksize, ts_out, h, L.ptr(out), L.cptr(cnt),
                       L.ptr(pair_ids), L.ptr(first), L.ptr(ws), ws.numel(), L.stream())
                n = int(L.read(cnt)[0])
                cs = CoordSet(out[:n].clone(), n, ts_out, ob)
                self._derived[key] = cs
                self._derived[("csr", ksize, ts_out)] = (first[:n + 1].clone(), pair_ids)
                return cs
            cap = max(self.n * K, 1)
            out = torch.empty(cap, dtype=torch.int64, device=dev)
            cnt = L.counter()
            nb = L.load().pcc_expand_ws_bytes(self.n, ksize)
            ws = L.workspace(nb, dev)
            L.call("pcc_coords_expand", L.ptr(self.keys), self.n, ksize, ts_out, ob.bit_mask(), L.ptr(out),
                   L.cptr(cnt), L.ptr(ws), ws.numel(), L.stream())
            n = int(L.read(cnt)[0])
            self._derived[key] = CoordSet(out[:n].clone(), n, ts_out, ob)
        return self._derived[key]

    def csr_map(self, ksize, ts_out):
        """(first, pair_ids) of the transposed map to expand(ksize, ts_out), or None when it was not built."""
        self.expand(ksize, ts_out)
        return self._derived.get(("csr", ksize, ts_out))

    # ---- kernel maps ---------------------------------------------------------------------------
    def kernel_map(self, out_set, ksize, transposed=False, up_stride=1, morton=None, step=None):
        """Map from this (input) set to `out_set` (a2-ii / a3), cached per (out set, kernel, kind).
        morton: visit the output rows in Z-curve order (default: for large stride-1 maps, see MORTON_MIN_ROWS)."""
        if morton is None:
            morton = (not transposed) and ksize > 1 and out_set.n >= MORTON_MIN_ROWS
        morton = bool(morton) and not transposed
        key = (out_set.uid, ksize, bool(transposed), up_stride, morton, step)
        m = self._maps.get(key)
        if m is not None:
            return m
        dev = self.device
        lib = L.load()
        K = ksize ** 3
        m = KernelMap()
        m.n_in, m.n_out, m.K, m.ksize, m.transposed = self.n, out_set.n, K, ksize, bool(transposed)
        m._pairs = None
        m.hdr = torch.empty(L.MAP_HDR_INTS, dtype=torch.int32, device=dev)
        nelem = lib.pcc_map_nbr_elems(out_set.n, ksize, up_stride, 1 if transposed else 0)
        m.nbr = torch.empty(max(nelem, 1), dtype=torch.int32, device=dev)
        m.rows = torch.empty(max(out_set.n, 1), dtype=torch.int32, device=dev) if (transposed or morton) else None
        m.d_pairs = torch.zeros(1, dtype=torch.int64, device=dev) if COUNT_PAIRS else None
        if step is None:
            step = out_set.ts if transposed else self.ts
        g = self.grid() if USE_GRID else None
        ws = L.workspace(lib.pcc_map_ws_bytes(out_set.n), dev)
        L.call("pcc_kernel_map_build", L.ptr(self.keys), self.n, L.ptr(out_set.keys), out_set.n, ksize, step,
               up_stride, 1 if transposed else 0, L.ptr(m.hdr), L.ptr(m.nbr), L.ptr(m.rows), L.ptr(m.d_pairs),
               L.ptr(g[0]) if g else None, L.ptr(g[1]) if g else None, g[2] if g else None,
               L.ptr(ws), ws.numel(), L.stream())
        self._maps[key] = m
        # (the cache key is the output set's serial number, not id(): nothing has to keep the output set alive, and a
        #  reference from here would close cycles such as y -> stride -> z -> expand -> children -> map onto y)
        return m


def _canon_check(keys, n):
    flag = L.counter(1, torch.int32)
    L.call("pcc_keys_is_canonical", L.ptr(keys), n, L.cptr(flag), L.stream())
    return bool(L.read(flag)[0])


def pack_keys(coords):
    """[n,4] int / float tensor on the GPU -> int64 keys (floor for floats), a1."""
    n = coords.shape[0]
    keys = torch.empty(max(n, 1), dtype=torch.int64, device=coords.device)
    if n == 0:
        return keys
    if coords.dtype.is_floating_point:
        c = coords.to(torch.float32).contiguous()
        L.call("pcc_keys_pack_f32", L.ptr(c), n, L.ptr(keys), L.stream())
    else:
        c = coords.to(torch.int32).contiguous()
        L.call("pcc_keys_pack_i32", L.ptr(c), n, L.ptr(keys), L.stream())
    return keys


def bounds_of(coords, canon_keys=None):
    """Bounds of user coordinates; with canon_keys (their packed keys) also whether those already are canonical, in the
    same device->host read: returns Bounds, or (Bounds, is_canonical)."""
    if coords.shape[0] == 0:
        b = Bounds(0, (0, 0, 0), (0, 0, 0))
        return b if canon_keys is None else (b, True)
    c = coords.contiguous()
    if c.dtype not in (torch.float32, torch.int32):
        c = c.to(torch.float32) if c.dtype.is_floating_point else c.to(torch.int32)
    out = torch.empty(12, dtype=torch.int32, device=c.device)
    L.call("pcc_coords_bounds", L.ptr(c), 1 if c.dtype.is_floating_point else 0, c.shape[0], L.ptr(out), L.stream())
    if canon_keys is not None:
        L.call("pcc_keys_is_canonical", L.ptr(canon_keys), c.shape[0], out.data_ptr() + 32, L.stream())
    v = out.tolist()                                      # one device->host read
    if v[0] < 0:
        raise L.PccError("negative batch index")
    b = Bounds(v[4], v[1:4], v[5:8])
    return b if canon_keys is None else (b, bool(v[8]))


class FrameRows:
    """Stand-in for the [n, 4] coordinate tensor of a frame whose keys / bounds / order flag `frame_intake` already
    produced: what `coordset_from_coords` reads of its argument (row count, device, the hint, the stride chain)."""

    def __init__(self, n, device, hint, chain=None):
        self.shape = (n, 4)
        self.device = device
        self._pcc_hint = hint
        self._pcc_chain = chain


def frame_intake(pc):
    """[n, 6] fp32 frame (x y z r g b) -> (keys int64 [n] of (0, floor xyz), features [n, 4] = (1, r, g, b), Bounds,
    canonical?) with one kernel and one host read (`pcc_frame_intake`; reference `model/model.py:141-161`)."""
    n = pc.shape[0]
    keys = torch.empty(n, dtype=torch.int64, device=pc.device)
    feats = torch.empty((n, 4), dtype=torch.float32, device=pc.device)
    out = torch.empty(12, dtype=torch.int32, device=pc.device)
    ws = L.workspace(L.load().pcc_frame_intake_ws_bytes(), pc.device)
    L.call("pcc_frame_intake", L.ptr(pc), n, L.ptr(keys), L.ptr(feats), L.ptr(out), L.ptr(ws), ws.numel(), L.stream())
    v = out.tolist()                                      # one device->host read
    return keys, feats, Bounds(0, v[1:4], [-x for x in v[5:8]]), bool(v[8])


def coords_intake(coords):
    """int32 [n, 4] coordinates (b, x, y, z) -> (keys, Bounds, canonical?) with one kernel pair and one read."""
    n = coords.shape[0]
    keys = torch.empty(n, dtype=torch.int64, device=coords.device)
    out = torch.empty(12, dtype=torch.int32, device=coords.device)
    ws = L.workspace(L.load().pcc_frame_intake_ws_bytes(), coords.device)
    L.call("pcc_coords_intake_i32", L.ptr(coords), n, L.ptr(keys), L.ptr(out), L.ptr(ws), ws.numel(), L.stream())
    v = out.tolist()                                      # one device->host read
    if v[0] < 0:
        raise L.PccError("negative batch index")
    return keys, Bounds(-v[4], v[1:4], [-x for x in v[5:8]]), bool(v[8])


def coordset_from_coords(coords, tensor_stride, stride_chain=None):
    """Canonicalise user coordinates.  Returns (CoordSet, perm, keep):
    perm  None when the rows already are in canonical order, else int64 [n] with
          canonical position -> row of the (de-duplicated) user-order tensor;
    keep  None, or int64 indices of the user rows that survive de-duplication (first wins, A.1).
    stride_chain: tensor strides of the sets the caller will derive next (`CoordSet.stride_chain_begin`); they are built
    from the same rows in the same batch of launches, and their sizes come back in the read this function makes anyway."""
    n = coords.shape[0]
    stride_chain = getattr(coords, "_pcc_chain", stride_chain)
    hint = getattr(coords, "_pcc_hint", None)             # (keys, bounds, canonical) already read by the caller (compress)
    if hint is not None:
        keys, b, canonical = hint[:3]
    elif n > 1 and coords.dtype == torch.int32 and coords.is_contiguous() and coords.data_ptr() % 16 == 0:
        keys, b, canonical = coords_intake(coords)
    else:
        keys = pack_keys(coords)
        b, canonical = bounds_of(coords, canon_keys=keys if n > 1 else None) if n > 1 else (bounds_of(coords), True)
    if n <= 1 or canonical:
        cs = CoordSet(keys, n, tensor_stride, b)
        if stride_chain and n > 1:
            resolve(cs.stride_chain_begin(stride_chain))
        return cs, None, None
    dev = coords.device
    lib = L.load()
    dims = [(b.hi[i] - b.lo[i]) // tensor_stride + 1 for i in range(3)]
    h = (C.c_int32 * 8)(b.lo[0], b.lo[1], b.lo[2], dims[0], dims[1], dims[2], tensor_stride, b.bmax + 1)
    words = lib.pcc_grid_words(h)
    on_lattice = all(v % tensor_stride == 0 for v in b.lo)      # rows of a stride-ts tensor sit on multiples of ts
    if USE_GRID and CANON_BY_GRID and on_lattice and words < (1 << 31) and words * 12 <= GRID_MAX_BYTES:
        # through the occupancy bitmap: canonical keys, first-wins row per coordinate and the grid index, no sort
        bits = torch.empty(words, dtype=torch.int64, device=dev)
        rank = torch.empty(words, dtype=torch.int32, device=dev)
        ukeys = torch.empty(n, dtype=torch.int64, device=dev)
        first_user = torch.empty(n, dtype=torch.int32, device=dev)
        cnt = L.counter(2)
        ws = L.workspace(lib.pcc_grid_ws_bytes(words), dev)
        L.call("pcc_keys_canonicalize_grid", L.ptr(keys), n, h, L.ptr(bits), L.ptr(rank), L.ptr(ukeys), L.ptr(first_user),
               L.cptr(cnt), L.ptr(ws), ws.numel(), L.stream())
        cs = CoordSet(ukeys, n, tensor_stride, b)          # (row count filled in below)
        # the strided sets of the chain do not need the canonical rows: marking coarse cells from the user rows is the same
        chain = cs.stride_chain_begin(stride_chain, keys=keys, n_keys=n) if stride_chain else None
        (nu, off_lattice), _ = resolve(Pending(cnt, lambda v: (int(v[0]), int(v[1]))), chain)
        if not off_lattice:
            cs.n = nu
            cs._grid = (bits, rank, h)
            cs._on_lattice = True
            first_user = first_user[:nu].long()
            if nu == n:
                return cs, first_user, None
            keep, inv = torch.sort(first_user)            # surviving user rows, original order
            rank_u = torch.empty_like(inv)
            rank_u[inv] = torch.arange(nu, device=dev)
            return cs, rank_u, keep
    skeys = torch.empty(n, dtype=torch.int64, device=dev)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    ws = L.workspace(lib.pcc_sort_ws_bytes(n), dev)
    L.call("pcc_sort_keys", L.ptr(keys), n, b.bit_mask(), L.ptr(skeys), L.ptr(perm), L.ptr(ws), ws.numel(), L.stream())
    ukeys = torch.empty(n, dtype=torch.int64, device=dev)
    first = torch.empty(n, dtype=torch.int32, device=dev)
    cnt = L.counter()
    ws = L.workspace(lib.pcc_unique_ws_bytes(n), dev)
    L.call("pcc_unique_sorted", L.ptr(skeys), n, L.ptr(ukeys), L.ptr(first), L.cptr(cnt), L.ptr(ws), ws.numel(),
           L.stream())
    nu = int(L.read(cnt)[0])
    # stable sort => first[] points at the smallest user row of each run
    first_user = perm.long()[first[:nu].long()]
    cs = CoordSet(ukeys[:nu].clone() if nu < n else ukeys, nu, tensor_stride, b)
    if nu == n:
        return cs, first_user, None
    keep, inv = torch.sort(first_user)            # surviving user rows, original order
    rank = torch.empty_like(inv)
    rank[inv] = torch.arange(nu, device=dev)
    return cs, rank, keep


# ------------------------------------------------------------------------------------------------
# functional operators on canonical-order features
# ------------------------------------------------------------------------------------------------
WEIGHT_OFFSET_ORDER = "x_fastest"   # enumeration of the K kernel offsets in `kernel[K, Cin, Cout]` of a state_dict


def weight_offset_perm(K, device):
    """SURVEY A.3 could not re-verify MinkowskiEngine's kernel-offset enumeration offline (x fastest is assumed:
    kidx = ix + k*iy + k*k*iz).  Should a checkpoint turn out to enumerate z fastest, set
    `sparse.WEIGHT_OFFSET_ORDER = "z_fastest"`: weights are then re-indexed while packing (and through autograd's
    index op in training); kernel maps, kernels and bitstreams are unaffected.  Returns None for the native order."""
    if WEIGHT_OFFSET_ORDER == "x_fastest" or K == 1:
        return None
    if WEIGHT_OFFSET_ORDER != "z_fastest":
        raise L.PccError(f"unknown WEIGHT_OFFSET_ORDER {WEIGHT_OFFSET_ORDER!r}")
    ks = round(K ** (1.0 / 3.0))
    idx = torch.arange(K, device=device)
    ix, iy, iz = idx % ks, (idx // ks) % ks, idx // (ks * ks)
    return iz + ks * iy + ks * ks * ix


class PackedConv:
    """Packed copy of a conv weight, refreshed when the parameter changes.  `transposed` selects the flattened
    [cin, K*cout] layout of the input-stationary generative transposed convolution."""

    def __init__(self, transposed=False):
        self.tag = None
        self.packed = None
        self.transposed = transposed

    def get(self, kernel, state_dict_order=False, tag_from=None):
        """state_dict_order: `kernel` is a module parameter as stored in a checkpoint (see WEIGHT_OFFSET_ORDER);
        otherwise its K axis already is in the native offset order.  tag_from: the Parameter `kernel` was derived from
        (e.g. `param[perm]`): a temporary's (data_ptr, _version) does not identify its contents -- the allocator
        recycles addresses -- so the cache is keyed on the parameter itself."""
        w = kernel.detach()
        src = tag_from if tag_from is not None else kernel
        tag = (src.data_ptr(), src._version, tuple(w.shape), str(w.device),
               WEIGHT_OFFSET_ORDER if (state_dict_order or tag_from is not None) else "")
        if tag != self.tag:
            w3 = w if w.dim() == 3 else w.unsqueeze(0)
            perm = weight_offset_perm(w3.shape[0], w3.device) if state_dict_order else None
            if perm is not None:
                w3 = w3[perm]
            w3 = w3.to(torch.float32).contiguous()
            K, cin, cout = w3.shape
            pre = "pcc_convt" if self.transposed else "pcc_conv"
            n = getattr(L.load(), pre + "_packed_elems")(K, cin, cout)
            if n <= 0:
                raise L.PccError(f"unsupported convolution shape K={K} cin={cin} cout={cout}")
            self.packed = torch.empty(n, dtype=torch.float32, device=w.device)
            L.call(pre + "_pack_weights", L.ptr(w3), K, cin, cout, L.ptr(self.packed), self.packed.numel(), L.stream())
            self.tag = tag
        return self.packed


def wants_pairs(K, cin, cout):
    """Does conv_forward try the pair-list form for this shape?  (The coordinate pre-passes of g_a / g_s queue the pair
    plans of such layers ahead of the features, so that their sizes are read together.)"""
    return bool(K >= PAIR_MIN_K and cin >= PAIR_MIN_CIN and L.load().pcc_conv_pairs_supported(K, cin, cout))


def conv_forward(feats, packed_w, bias, K, cin, cout, kmap, n_out, act=L.ACT_NONE, slope=0.01):
    """out[o] = act(bias + sum_k feats[nbr_k(o)] @ W[k])  -- a2-iii / a3."""
    feats = feats.contiguous()
    out = torch.empty((n_out, cout), dtype=torch.float32, device=feats.device)
    if n_out == 0:
        return out
    b = bias.detach().reshape(-1).contiguous() if bias is not None else None
    if kmap is not None and wants_pairs(K, cin, cout):
        plan = kmap.pair_plan()
        if plan is not None:           # sparse map: gathered GEMM over the compacted pairs, then ordered reduce
            pos, pair_in, tile_k, info, padded = plan
            T = L.workspace(max(padded, 1) * cout * 4, feats.device)
            L.call("pcc_conv_fwd_pairs", L.ptr(feats), feats.shape[0], cin, L.ptr(packed_w), L.ptr(b), K, cout,
                   L.ptr(pair_in), L.ptr(tile_k), L.ptr(info), padded, L.ptr(pos), n_out, L.ptr(T), L.ptr(out), act,
                   float(slope), *L.arith_args(feats.device), L.stream())
            return out
    ws = L.workspace(L.load().pcc_conv_ws_bytes(feats.shape[0], K, cin, cout), feats.device)
    L.call("pcc_conv_fwd", L.ptr(feats), feats.shape[0], cin, L.ptr(packed_w), L.ptr(b), K, cout,
           L.ptr(kmap.hdr) if kmap is not None else None, L.ptr(kmap.nbr) if kmap is not None else None,
           L.ptr(kmap.rows) if kmap is not None else None, n_out, L.ptr(out), act, float(slope), L.ptr(ws),
           ws.numel(), *L.arith_args(feats.device), L.stream())
    return out


def conv_head_forward(feats, packed_w0, bias0, cmid, w2, bias2, cset, kmap):
    """Occupancy head over a canonical set: conv_k3(relu(conv_k3(x))) -> [n, 1] logits, hidden features never stored
    (`predict_i`, reference `model/transforms.py:141-160`).  w2: the second kernel in its packed thin layout."""
    feats = feats.contiguous()
    n, cin = feats.shape
    out = torch.empty((n, 1), dtype=torch.float32, device=feats.device)
    if n == 0:
        return out
    bands = cset.band_tiles()
    ws = L.workspace(L.load().pcc_conv_head_ws_bytes(n), feats.device)
    b0 = bias0.detach().reshape(-1).contiguous() if bias0 is not None else None
    b2 = bias2.detach().reshape(-1).contiguous() if bias2 is not None else None
    L.call("pcc_conv_head_fwd", L.ptr(feats), n, cin, L.ptr(packed_w0), L.ptr(b0), cmid, L.ptr(w2), L.ptr(b2),
           L.ptr(kmap.hdr), L.ptr(kmap.nbr), L.ptr(bands[0]) if bands else None, L.ptr(bands[1]) if bands else None,
           L.ptr(out), L.ptr(ws), ws.numel(), L.stream())
    return out


def convt_forward(feats, packed_w, bias, K, cin, cout, kmap, n_out, act=L.ACT_NONE, slope=0.01):
    """Generative transposed conv (a3): dense GEMM into the per-pair buffer T, then ordered gather-sum."""
    feats = feats.contiguous()
    n_in = feats.shape[0]
    out = torch.empty((n_out, cout), dtype=torch.float32, device=feats.device)
    if n_out == 0 or n_in == 0:
        return out
    T = torch.empty(n_in * K * cout, dtype=torch.float32, device=feats.device)
    b = bias.detach().reshape(-1).contiguous() if bias is not None else None
    L.call("pcc_convt_fwd", L.ptr(feats), n_in, cin, L.ptr(packed_w), L.ptr(b), K, cout, L.ptr(kmap.hdr),
           L.ptr(kmap.nbr), L.ptr(kmap.rows), n_out, L.ptr(T), L.ptr(out), act, float(slope), *L.arith_args(feats.device),
           L.stream())
    return out


def convt_forward_csr(feats, packed_w, bias, K, cin, cout, csr, n_out, act=L.ACT_NONE, slope=0.01, ex_map=None,
                      ex_bias=None):
    """Generative transposed conv (a3) with the CSR pair lists produced by the coordinate expansion.
    ex_map / ex_bias: a conv map of the output set and a [K2, cout] table; row o additionally receives ex_bias[k] for
    every neighbour k it has (the constant part of two fused affine layers)."""
    feats = feats.contiguous()
    n_in = feats.shape[0]
    out = torch.empty((n_out, cout), dtype=torch.float32, device=feats.device)
    if n_out == 0 or n_in == 0:
        return out
    first, pair_ids = csr
    T = torch.empty(n_in * K * cout, dtype=torch.float32, device=feats.device)
    b = bias.detach().reshape(-1).contiguous() if bias is not None else None
    eb = ex_bias.detach().to(torch.float32).contiguous() if ex_map is not None else None
    L.call("pcc_convt_fwd_csr", L.ptr(feats), n_in, cin, L.ptr(packed_w), L.ptr(b), K, cout, L.ptr(first),
           L.ptr(pair_ids), n_out, L.ptr(T), L.ptr(out), act, float(slope),
           L.ptr(ex_map.nbr) if ex_map is not None else None, ex_map.K if ex_map is not None else 0, L.ptr(eb),
           *L.arith_args(feats.device), L.stream())
    return out


def csr_pair_total(csr, n_out):
    """Number of pairs of CSR lists (host read; accounting / tests only): first[n_out] of the prefix form, the slots' fill of the
    slotted form (`csr_for(slots=True)`)."""
    if len(csr) > 2:
        wg_end = csr[2]
        slot = csr[1].numel() // wg_end.numel()
        return int((wg_end.long() - torch.arange(wg_end.numel(), device=wg_end.device) * slot).sum().item())
    return int(csr[0][n_out].item())


def convt_forward_csr_grid(feats, packed_w, bias, K, cin, cout, csr, out_set, act, ex_bias, slope=0.01):
    """`convt_forward_csr` with the constant-per-existing-neighbour term keyed on the output set's own grid index
    (no 3x3x3 kernel map of the candidate set)."""
    feats = feats.contiguous()
    n_in, n_out = feats.shape[0], out_set.n
    out = torch.empty((n_out, cout), dtype=torch.float32, device=feats.device)
    if n_out == 0 or n_in == 0:
        return out
    first, pair_ids = csr[0], csr[1]
    wg_end = csr[2] if len(csr) > 2 else None                  # slotted lists (`csr_for(slots=True)`)
    g = out_set.grid()
    T = torch.empty(n_in * K * cout, dtype=torch.float32, device=feats.device)
    b = bias.detach().reshape(-1).contiguous() if bias is not None else None
    L.call("pcc_convt_fwd_csr_grid", L.ptr(feats), n_in, cin, L.ptr(packed_w), L.ptr(b), K, cout, L.ptr(first), L.ptr(pair_ids),
           n_out, L.ptr(T), L.ptr(out), act, float(slope), L.ptr(out_set.keys), L.ptr(g[0]), L.ptr(g[1]), g[2],
           L.ptr(ex_bias.detach().to(torch.float32).contiguous()), L.ptr(wg_end), *L.arith_args(feats.device), L.stream())
    return out


def convt_forward_csr_chunked(feats, packed_w, bias, K, cin, cout, csr, in_set, out_set, act, ex_bias=None, slope=0.01):
    """`convt_forward_csr` / `convt_forward_csr_grid` without the whole per-pair buffer: parent rows go through in chunks
    whose products fit the Infinity Cache (staging buffer in the shared workspace), partial sums carried in the output."""
    feats = feats.contiguous()
    n_in, n_out = feats.shape[0], out_set.n
    out = torch.empty((n_out, cout), dtype=torch.float32, device=feats.device)
    if n_out == 0 or n_in == 0:
        return out
    first, pair_ids = csr
    lib = L.load()
    tb, wb = lib.pcc_convt_chunk_t_bytes(n_in, K, cout), lib.pcc_convt_chunk_ws_bytes(n_in, K, cout)
    ws = L.workspace(tb + wb, feats.device)
    g = out_set.grid() if ex_bias is not None else None
    if ex_bias is not None and g is None:
        raise L.PccError("convt_forward_csr_chunked: ex_bias needs the grid index of the output set")
    b = bias.detach().reshape(-1).contiguous() if bias is not None else None
    eb = ex_bias.detach().to(torch.float32).contiguous() if ex_bias is not None else None
    L.call("pcc_convt_fwd_csr_chunked", L.ptr(feats), n_in, cin, L.ptr(packed_w), L.ptr(b), K, cout, L.ptr(first), L.ptr(pair_ids),
           n_out, L.ptr(in_set.keys), L.ptr(out_set.keys), out_set.ts, ws.data_ptr() + wb, tb, L.ptr(out), act, float(slope),
           L.ptr(g[0]) if g else None, L.ptr(g[1]) if g else None, g[2] if g else None, L.ptr(eb), ws.data_ptr(), wb,
           *L.arith_args(feats.device), L.stream())
    return out


def conv_thin_grid_forward(feats, packed_w, bias, cin, cout, cset):
    """3x3x3 convolution to <= 4 channels over a full set, neighbour rows from the set's grid index."""
    feats = feats.contiguous()
    n = feats.shape[0]
    out = torch.empty((n, cout), dtype=torch.float32, device=feats.device)
    if n == 0:
        return out
    g = cset.grid()
    ws = L.workspace(L.load().pcc_thin_grid_ws_bytes(n, cout), feats.device)
    b = bias.detach().reshape(-1).contiguous() if bias is not None else None
    L.call("pcc_conv_thin_grid_fwd", L.ptr(feats), n, cin, L.ptr(packed_w), L.ptr(b), cout, L.ptr(cset.keys), L.ptr(g[0]), L.ptr(g[1]),
           g[2], L.ptr(out), L.ptr(ws), ws.numel(), L.stream())
    return out


def convt_forward_rows(feats, packed_w, bias, K, cin, cout, csr, n_out, act=L.ACT_NONE, slope=0.01, pairs_bound=None):
    """Transposed conv evaluated for the `n_out` output rows whose CSR pair lists are given (work ~ their pairs)."""
    feats = feats.contiguous()
    out = torch.empty((n_out, cout), dtype=torch.float32, device=feats.device)
    if n_out == 0:
        return out
    first, pair_ids = csr
    lib = L.load()
    # (an upper bound on the CSR entries would save this read, but the pair GEMM's grid, its padding memset and the
    #  bucket scan are sized by it: measured +1 ms per step with the bound ceil(k/2)^3 per row -- the exact count stays)
    pairs = int(first[n_out].item()) if pairs_bound is None else pairs_bound
    T = torch.empty(max(lib.pcc_convt_rows_t_elems(pairs, K, cout), 1), dtype=torch.float32, device=feats.device)
    ws = L.workspace(lib.pcc_convt_rows_int_ws_bytes(pairs, K), feats.device)
    b = bias.detach().reshape(-1).contiguous() if bias is not None else None
    L.call("pcc_convt_fwd_rows", L.ptr(feats), feats.shape[0], cin, L.ptr(packed_w), L.ptr(b), K, cout, L.ptr(first),
           L.ptr(pair_ids), n_out, pairs, L.ptr(T), L.ptr(out), act, float(slope), L.ptr(ws), ws.numel(),
           *L.arith_args(feats.device), L.stream())
    return out


def map_from_csr(csr, n_in, n_out, ksize):
    """Conv-form KernelMap (nbr[k][o] = input row) of a transposed conv restricted to `n_out` output rows, from their
    CSR pair lists: conv_forward then evaluates it (pair-list form when sparse) without the dense per-pair buffer."""
    first, pair_ids = csr
    dev = first.device
    K = ksize ** 3
    m = KernelMap()
    m.n_in, m.n_out, m.K, m.ksize, m.transposed = n_in, n_out, K, ksize, False
    m._pairs, m.d_pairs, m.rows = None, None, None
    m.hdr = torch.empty(L.MAP_HDR_INTS, dtype=torch.int32, device=dev)
    m.nbr = torch.empty(max(K * n_out, 1), dtype=torch.int32, device=dev)
    if n_out:
        L.call("pcc_map_from_csr", L.ptr(first), L.ptr(pair_ids), n_out, ksize, L.ptr(m.hdr), L.ptr(m.nbr), L.stream())
    return m


def conv_wgrad(feats_in, grad_out, K, cin, cout, kmap):
    """dW[k][ci][co] = sum over pairs of offset k of feats_in[i][ci] * grad_out[o][co]  (kmap None: K = 1 identity)."""
    feats_in, grad_out = feats_in.contiguous(), grad_out.contiguous()
    dW = torch.empty((K, cin, cout), dtype=torch.float32, device=feats_in.device)
    ws = L.workspace(L.load().pcc_conv_wgrad_ws_bytes(grad_out.shape[0], K, cin, cout), feats_in.device)
    L.call("pcc_conv_wgrad", L.ptr(feats_in), feats_in.shape[0], cin, L.ptr(grad_out), grad_out.shape[0], cout, K,
           L.ptr(kmap.hdr) if kmap is not None else None, L.ptr(kmap.nbr) if kmap is not None else None,
           L.ptr(kmap.rows) if kmap is not None else None, L.ptr(dW), L.ptr(ws), ws.numel(), L.stream())
    return dW


def conv_wgrad_self(feats, grad_out, K, cin, kmap, cout=1):
    """dW [K, cin, cout] (cout 1 or 16) of an odd-kernel convolution over a set mapped onto itself (`kmap` = that self map):
    input-stationary, dW[k][ci][co] = sum_i feats[i][ci] * grad_out[nbr_{K-1-k}(i)][co]."""
    feats, grad_out = feats.contiguous(), grad_out.contiguous()
    n = feats.shape[0]
    dW = torch.empty((K, cin, cout), dtype=torch.float32, device=feats.device)
    ws = L.workspace(L.load().pcc_conv_wgrad_self_ws_bytes(n, K, cin, cout), feats.device)
    L.call("pcc_conv_wgrad_self", L.ptr(feats), n, cin, L.ptr(grad_out), cout, K, L.ptr(kmap.hdr), L.ptr(kmap.nbr), L.ptr(dW),
           L.ptr(ws), ws.numel(), L.stream())
    return dW


def convt_scatter_rows(grad_out, csr, n_pairs, cout):
    """dT[pair] = grad_out[output row of the pair] for the input-stationary transposed conv."""
    first, pair_ids = csr
    grad_out = grad_out.contiguous()
    dT = torch.empty((n_pairs, cout), dtype=torch.float32, device=grad_out.device)
    L.call("pcc_convt_scatter_rows", L.ptr(grad_out), L.ptr(first), L.ptr(pair_ids), grad_out.shape[0], cout, L.ptr(dT),
           L.stream())
    return dT


def topk_mask(logits, seg_begin, ks):
    """Per-batch top-k mask over rows in canonical order (a4)."""
    n = logits.shape[0]
    mask = torch.empty(max(n, 1), dtype=torch.uint8, device=logits.device)
    if n == 0:
        return mask[:0].bool()
    nb = len(ks)
    sb = (C.c_int64 * (nb + 1))(*seg_begin)
    kk = (C.c_int64 * nb)(*[int(k) for k in ks])
    ws = L.workspace(L.load().pcc_topk_ws_bytes(n), logits.device)
    if logits.dim() == 2:     # column 0 of an [n,c] logit tensor (`prediction.F[:, 0]`, model/transforms.py:246)
        if logits.stride(1) != 1 and logits.shape[1] != 1:
            logits = logits.contiguous()
        stride = logits.stride(0)
    else:
        logits, stride = logits.contiguous(), 1
    if not logits.is_cuda:
        raise L.PccError("topk_mask: GPU tensor required")
    L.call("pcc_topk_mask", logits.data_ptr(), stride, sb, kk, nb, L.ptr(mask), L.ptr(ws), ws.numel(), L.stream())
    return mask[:n].view(torch.bool)


def topk_prune_keys(logits, seg_begin, ks, keys):
    """`topk_mask` and the compaction of the kept rows' keys in one pass over the logits (a4, coordinates only: what the
    decoder's composite levels need).  Returns (mask bool [n], kept keys int64 [n_keep], n_keep)."""
    n = logits.shape[0]
    nb = len(ks)
    n_keep = sum(min(max(int(k), 0), seg_begin[i + 1] - seg_begin[i]) for i, k in enumerate(ks))
    mask = torch.empty(max(n, 1), dtype=torch.uint8, device=logits.device)
    keys_out = torch.empty(max(n_keep, 1), dtype=torch.int64, device=logits.device)
    if n == 0:
        return mask[:0].bool(), keys_out[:0], 0
    if logits.dim() == 2:
        if logits.stride(1) != 1 and logits.shape[1] != 1:
            logits = logits.contiguous()
        stride = logits.stride(0)
    else:
        logits, stride = logits.contiguous(), 1
    if not logits.is_cuda:
        raise L.PccError("topk_prune_keys: GPU tensor required")
    sb = (C.c_int64 * (nb + 1))(*seg_begin)
    kk = (C.c_int64 * nb)(*[int(k) for k in ks])
    ws = L.workspace(L.load().pcc_topk_ws_bytes(n), logits.device)
    L.call("pcc_topk_prune_keys", logits.data_ptr(), stride, sb, kk, nb, L.ptr(keys), L.ptr(mask), L.ptr(keys_out), L.ptr(ws),
           ws.numel(), L.stream())
    return mask[:n].view(torch.bool), keys_out[:n_keep], n_keep


def prune(keys, n, feats, mask, n_keep=None):
    """Stable row compaction (a4).  n_keep known (top-k) avoids the device->host count read."""
    dev = keys.device
    m = mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
    m = m.contiguous()
    c = feats.shape[1] if feats is not None else 0
    keys_out = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    feat_out = torch.empty((max(n, 1), c), dtype=torch.float32, device=dev) if feats is not None else None
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = L.workspace(L.load().pcc_prune_ws_bytes(n), dev)
    f = feats.contiguous() if feats is not None else None
    L.call("pcc_prune_rows", L.ptr(m), n, L.ptr(keys), L.ptr(f), c, L.ptr(keys_out), L.ptr(feat_out), L.ptr(cnt),
           L.ptr(ws), ws.numel(), L.stream())
    k = int(cnt.item()) if n_keep is None else int(n_keep)
    return keys_out[:k], (feat_out[:k] if feat_out is not None else None), k


def lookup_gather(cset, feats, query_keys, nq):
    """features_at_coordinates for on-grid queries (a6)."""
    feats = feats.contiguous()
    out = torch.empty((nq, feats.shape[1]), dtype=torch.float32, device=feats.device)
    if nq:
        L.call("pcc_lookup_gather", L.ptr(cset.keys), cset.n, L.ptr(feats), feats.shape[1], L.ptr(query_keys), nq,
               L.ptr(out), L.stream())
    return out
