"""ctypes binding of libpcc_hip.so (the C ABI declared in include/pcc_hip.h).

PyTorch-ROCm supplies device memory (`tensor.data_ptr()`) and the HIP stream; every kernel on the
hot path lives in the shared library.  There is NO CPU fallback: if the library is missing or a
call fails, this module raises.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpcc_hip.so")

MAP_HDR_INTS = 512
FORM_NAMES = ("other", "k_gemm_h2", "k_gemm_bf2", "k_pair_h2", "pair_bf", "k_conv_mfma_bf", "k_conv_mfma", "k_conv_wave16",
              "k_convt_gather_csr")
MAP_MAX_SEG = 8
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2

_p = C.c_void_p
_i32, _i64, _u64, _f32, _sz = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/pcc_hip.h one to one
SIGNATURES = {
    "pcc_version": (C.c_int, []),
    "pcc_last_error": (C.c_char_p, []),
    "pcc_device_info": (C.c_int, [C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "pcc_coords_bounds": (C.c_int, [_p, _i32, _i64, _p, _p]),
    "pcc_rows_gather": (C.c_int, [_p, _p, _i64, _i32, _p, _p]),
    "pcc_keys_pack_i32": (C.c_int, [_p, _i64, _p, _p]),
    "pcc_keys_pack_f32": (C.c_int, [_p, _i64, _p, _p]),
    "pcc_keys_unpack": (C.c_int, [_p, _i64, _p, _p]),
    "pcc_batch_bounds": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p]),
    "pcc_sort_ws_bytes": (_sz, [_i64]),
    "pcc_sort_keys": (C.c_int, [_p, _i64, _u64, _p, _p, _p, _sz, _p]),
    "pcc_unique_ws_bytes": (_sz, [_i64]),
    "pcc_unique_sorted": (C.c_int, [_p, _i64, _p, _p, _p, _p, _sz, _p]),
    "pcc_keys_is_canonical": (C.c_int, [_p, _i64, _p, _p]),
    "pcc_stride_ws_bytes": (_sz, [_i64]),
    "pcc_coords_stride": (C.c_int, [_p, _i64, _i32, _u64, _p, _p, _p, _sz, _p]),
    "pcc_expand_ws_bytes": (_sz, [_i64, _i32]),
    "pcc_coords_expand": (C.c_int, [_p, _i64, _i32, _i32, _u64, _p, _p, _p, _sz, _p]),
    "pcc_expand_csr_ws_bytes": (_sz, [_i64, _i32]),
    "pcc_coords_expand_csr": (C.c_int, [_p, _i64, _i32, _i32, C.POINTER(_i32), _p, _p, _p, _p, _p, _sz, _p]),
    "pcc_map_nbr_elems": (_i64, [_i64, _i32, _i32, _i32]),
    "pcc_map_ws_bytes": (_sz, [_i64]),
    "pcc_kernel_map_build": (C.c_int, [_p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p,
                                       C.POINTER(_i32), _p, _sz, _p]),
    "pcc_grid_words": (_i64, [C.POINTER(_i32)]),
    "pcc_grid_ws_bytes": (_sz, [_i64]),
    "pcc_grid_build": (C.c_int, [_p, _i64, C.POINTER(_i32), _p, _p, _p, _sz, _p]),
    "pcc_coords_stride_grid": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "pcc_keys_canonicalize_grid": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "pcc_coords_expand_grid": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "pcc_expand_grid_csr_ws_bytes": (_sz, [_i64]),
    "pcc_coords_expand_grid_csr": (C.c_int, [_p, _i64, _i32, _i32, _p, _p, _p, _i64, _p, _p, _p, _p, _sz, _p]),
    "pcc_coords_expand_grid_csr_zk": (C.c_int, [_p, _i64, _i32, _i32, _p, _p, _p, _i64, _p, _p, _p, _p, _sz, _p]),
    "pcc_expand_grid_csr_slot_elems": (_i64, [_i64, _i32]),
    "pcc_coords_expand_grid_csr_slots": (C.c_int, [_p, _i64, _i32, _i32, _p, _p, _p, _i64, _p, _p, _p, _p, _i32, _p]),
    "pcc_map_to_dense": (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p]),
    "pcc_set_in4_min_rows": (C.c_int, [_i64]),
    "pcc_set_thin_z_min_rows": (C.c_int, [_i64]),
    "pcc_conv_packed_elems": (_i64, [_i32, _i32, _i32]),
    "pcc_conv_pack_weights": (C.c_int, [_p, _i32, _i32, _i32, _p, _i64, _p]),
    "pcc_conv_pack_weights_ex": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, _p, _i64, _p]),
    "pcc_conv_ws_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "pcc_conv_fwd": (C.c_int, [_p, _i64, _i32, _p, _p, _i32, _i32, _p, _p, _p, _i64, _p, _i32, _f32, _p, _sz, _i32, _p, _p]),
    "pcc_conv_head_supported": (C.c_int, [_i32, _i32]),
    "pcc_conv_head_ws_bytes": (_sz, [_i64]),
    "pcc_conv_head_fwd": (C.c_int, [_p, _i64, _i32, _p, _p, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "pcc_band_tiles_cap": (_i64, [_i64, _i32, _i32]),
    "pcc_band_tiles_ws_bytes": (_sz, [_i32, _i32]),
    "pcc_band_tiles_build": (C.c_int, [_p, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _p, _i64, _p, _p, _sz, _p]),
    "pcc_convt_packed_elems": (_i64, [_i32, _i32, _i32]),
    "pcc_convt_pack_weights": (C.c_int, [_p, _i32, _i32, _i32, _p, _i64, _p]),
    "pcc_convt_fwd": (C.c_int, [_p, _i64, _i32, _p, _p, _i32, _i32, _p, _p, _p, _i64, _p, _p, _i32, _f32, _i32, _p, _p]),
    "pcc_convt_fwd_csr": (C.c_int, [_p, _i64, _i32, _p, _p, _i32, _i32, _p, _p, _i64, _p, _p, _i32, _f32, _p, _i32, _p,
                                    _i32, _p, _p]),
    "pcc_convt_rows_int_ws_bytes": (_sz, [_i64, _i32]),
    "pcc_convt_rows_t_elems": (_i64, [_i64, _i32, _i32]),
    "pcc_convt_fwd_rows": (C.c_int, [_p, _i64, _i32, _p, _p, _i32, _i32, _p, _p, _i64, _i64, _p, _p, _i32, _f32, _p, _sz,
                                     _i32, _p, _p]),
    "pcc_map_from_csr": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p]),
    "pcc_conv_wgrad_ws_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "pcc_conv_wgrad": (C.c_int, [_p, _i64, _i32, _p, _i64, _i32, _i32, _p, _p, _p, _p, _p, _sz, _p]),
    "pcc_conv_wgrad_self_supported": (C.c_int, [_i32, _i32, _i32]),
    "pcc_gdn_bwd_pre": (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p, _p]),
    "pcc_gdn_bwd_post": (C.c_int, [_p, _p, _p, _i64, _p]),
    "pcc_gdn_gamma_eff": (C.c_int, [_p, _i32, _p, _p]),
    "pcc_gdn_reparam_bwd": (C.c_int, [_p, _p, _p, _p, _i32, C.c_float, _p, _p, _p]),
    "pcc_focal_rows": (C.c_int, [_p, _i64, _p, _p, _i64, _p, _i32, C.c_float, C.c_float, _p, _p, _p]),
    "pcc_quant_mlp_params": (_i32, []),
    "pcc_quant_mlp_ws_bytes": (_sz, [_i64]),
    "pcc_quant_mlp_fwd": (C.c_int, [_p, _p, _i64, _p, _p, _p]),
    "pcc_quant_mlp_bwd": (C.c_int, [_p, _p, _p, _i64, _p, _p, _p, _p, _p, _sz, _p]),
    "pcc_conv_wgrad_self_ws_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "pcc_conv_wgrad_self": (C.c_int, [_p, _i64, _i32, _p, _i32, _i32, _p, _p, _p, _p, _sz, _p]),
    "pcc_convt_scatter_rows": (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p]),
    "pcc_convt_fwd_csr_grid": (C.c_int, [_p, _i64, _i32, _p, _p, _i32, _i32, _p, _p, _i64, _p, _p, _i32, _f32, _p, _p, _p, C.POINTER(_i32), _p, _p, _i32, _p, _p]),
    "pcc_set_t_chunk_bytes": (C.c_int, [_i64]),
    "pcc_convt_chunk_t_bytes": (_sz, [_i64, _i32, _i32]),
    "pcc_convt_chunk_ws_bytes": (_sz, [_i64, _i32, _i32]),
    "pcc_convt_fwd_csr_chunked": (C.c_int, [_p, _i64, _i32, _p, _p, _i32, _i32, _p, _p, _i64, _p, _p, _i32, _p, _sz, _p, _i32, _f32,
                                            _p, _p, C.POINTER(_i32), _p, _p, _sz, _i32, _p, _p]),
    "pcc_thin_grid_ws_bytes": (_sz, [_i64, _i32]),
    "pcc_conv_thin_grid_fwd": (C.c_int, [_p, _i64, _i32, _p, _p, _i32, _p, _p, _p, C.POINTER(_i32), _p, _p, _sz, _p]),
    "pcc_gauss_lik_fwd": (C.c_int, [_p, _p, _p, _i64, _p, _p]),
    "pcc_gauss_lik_bwd": (C.c_int, [_p, _p, _p, _p, _i64, _p, _p, _p, _p]),
    "pcc_gdn_packed_elems": (_i64, [_i32]),
    "pcc_gdn_pack": (C.c_int, [_p, _p, _i32, _f32, _p, _i64, _p, _p]),
    "pcc_gdn_fwd": (C.c_int, [_p, _i64, _i32, _p, _p, _i32, _p, _i32, _p]),
    "pcc_frame_intake_ws_bytes": (_sz, []),
    "pcc_coords_intake_i32": (C.c_int, [_p, _i64, _p, _p, _p, _sz, _p]),
    "pcc_frame_intake": (C.c_int, [_p, _i64, _p, _p, _p, _p, _sz, _p]),
    "pcc_decode_finish": (C.c_int, [_p, _p, _i64, _p, _p]),
    "pcc_topk_ws_bytes": (_sz, [_i64]),
    "pcc_topk_mask": (C.c_int, [_p, _i64, C.POINTER(_i64), C.POINTER(_i64), _i32, _p, _p, _sz, _p]),
    "pcc_topk_prune_keys": (C.c_int, [_p, _i64, C.POINTER(_i64), C.POINTER(_i64), _i32, _p, _p, _p, _p, _sz, _p]),
    "pcc_prune_ws_bytes": (_sz, [_i64]),
    "pcc_prune_rows": (C.c_int, [_p, _i64, _p, _p, _i32, _p, _p, _p, _p, _sz, _p]),
    "pcc_lookup_gather": (C.c_int, [_p, _i64, _p, _i32, _p, _i64, _p, _p]),
    "pcc_lookup_rows": (C.c_int, [_p, _i64, _p, _i64, _p, _p]),
    "pcc_gauss_encode": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _i32, _p, _p, _p, _p]),
    "pcc_gauss_decode": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _i32, _p, _p, _p]),
    "pcc_eb_encode": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _p, _p, _p]),
    "pcc_eb_lik_fwd": (C.c_int, [_p, _i64, _i32, _p, _p, _p]),
    "pcc_eb_lik_bwd": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p]),
    "pcc_pmf_to_quantized_cdf": (C.c_int, [_p, _i32, _i32, _p]),
    "pcc_rans_max_bytes": (_i64, [_i64]),
    "pcc_rans_encode_host": (C.c_int, [_p, _p, _i64, _p, _i32, _p, _p, _p, _i64, C.POINTER(_i64)]),
    "pcc_rans_decode_host": (C.c_int, [_p, _i64, _p, _i64, _p, _i32, _p, _p, _p]),
    "pcc_rans_container_max_bytes": (_i64, [_i64, _i32]),
    "pcc_rans_streams_ws_bytes": (_sz, [_i64, _i32]),
    "pcc_rans_build_enc_table": (C.c_int, [_p, _i32, _i32, _p, _p]),
    "pcc_rans_stream_symbols": (_i64, [_i64, _i32, _i32, _i32]),
    "pcc_rans_estimate_bits": (C.c_int, [_p, _p, _i64, _i32, _p, _i32, _p, _p, _p, _p]),
    "pcc_rans_encode_streams": (C.c_int, [_p, _p, _i64, _i32, _i32, _i32, _p, _i32, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "pcc_rans_dec_table_bytes": (_i64, [_i32, _p]),
    "pcc_rans_build_dec_table": (C.c_int, [_p, _i32, _i32, _p, _p]),
    "pcc_rans_decode_streams": (C.c_int, [_p, _i64, _p, _i64, _i32, _i32, _i32, _p, _i32, _p, _p, _p, _i64, _p, _p, _p]),
    "pcc_octree_max_bytes": (_i64, [_i64, _i32]),
    "pcc_octree_encode_host": (C.c_int, [_p, _i64, _i32, _p, _i64, C.POINTER(_i64)]),
    "pcc_octree_decode_host": (C.c_int, [_p, _i64, _p, _i64, C.POINTER(_i64), C.POINTER(_i32)]),
    "pcc_conv_pairs_supported": (C.c_int, [_i32, _i32, _i32]),
    "pcc_pair_plan_ws_bytes": (_sz, [_i64, _i32]),
    "pcc_pair_plan_rank": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _p, _sz, _p]),
    "pcc_pair_plan_fill": (C.c_int, [_p, _p, _p, _i64, _i32, _i64, _p, _p, _p]),
    "pcc_conv_fwd_pairs": (C.c_int, [_p, _i64, _i32, _p, _p, _i32, _i32, _p, _p, _p, _i64, _p, _i64, _p, _p, _i32,
                                     C.c_float, _i32, _p, _p]),
    "pcc_nn_sorted_x": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _p]),
    "pcc_prof_enable": (C.c_int, [_i32]),
    "pcc_prof_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(_i64)]),
    "pcc_prof_sequence": (_i64, [C.POINTER(_i32), _i64]),
    "pcc_prof_collect_forms": (C.c_int, [C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


class PccError(RuntimeError):
    pass


def load():
    """Load libpcc_hip.so and declare every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PccError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C unified_point_cloud_compression_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().pcc_last_error()
        raise PccError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous and on a GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise PccError("libpcc_hip operates on GPU tensors only (no CPU fallback); got a CPU tensor")
    if not t.is_contiguous():
        raise PccError("non-contiguous tensor handed to libpcc_hip")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """The current HIP stream of the current device as an integer handle.  (`torch.cuda.current_stream()` builds a Stream
    object through four Python frames -- 8 us per call, 100 calls per step; the raw query is the same handle in 0.2 us.)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_side = {}


def side_stream(device, which=0):
    """Extra HIP streams per device for work that is independent of what the main stream is doing: 0 = latency-bound kernels on a
    handful of CUs (the hyper-latent's rANS decode beside the decoder's coordinate work), 1 = host->device uploads (a pageable
    copy waits for everything queued before it on ITS stream, so it must not share one with a long kernel).  Tensors allocated
    under a side stream must be `record_stream`-ed on the stream that consumes them."""
    key = (torch.device(device).index or 0, which)
    st = _side.get(key)
    if st is None:
        st = _side[key] = torch.cuda.Stream(device=device)
    return st


_ws = {}


def workspace(nbytes, device):
    """One growing scratch buffer per (device, current stream); calls on a stream are ordered, so they share it -- and work on a
    side stream gets a buffer of its own instead of racing the main stream's."""
    key = (torch.device(device).index or 0, stream())
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        _ws[key] = buf = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
    return buf


# ---- zero-copy counters --------------------------------------------------------------------------------------
# Sizes the host needs next (rows of a derived coordinate set, a canonical-order flag, bytes of a bitstream) are written
# by the kernels straight into pinned host memory (device-visible at the same address under HIP's unified addressing):
# the host zeroes the slot, launches, synchronises the stream and reads -- no device allocation, no fill kernel and no
# device->host copy launch per read (round 1: ~40 `.item()` reads per step, 25-50 us of idle GPU each).
_pin = {"buf": None, "next": 0, "dev_blocks": {}}
_PIN_SLOTS, _PIN_WORDS = 1024, 4
PINNED_COUNTERS = os.environ.get("PCC_PINNED_COUNTERS", "0") != "0"   # measured 0.3 ms/step slower than `.item()` reads (round 2): off


def counter(n=1, dtype=torch.int64):
    """Zeroed pinned host tensor of `n` (<= 4) int64 (or 2n int32) words for a kernel to write into."""
    if not PINNED_COUNTERS:           # default: device counters cut from a zeroed block (one fill per 1024 counters instead of
        d = torch.cuda.current_device()          # one per counter), read back with a copy; one block per device
        st = _pin["dev_blocks"].setdefault(d, {"blk": None, "next": 0})
        blk = st["blk"]
        if blk is None or st["next"] + _PIN_WORDS > blk.numel():
            blk = st["blk"] = torch.zeros(1024 * _PIN_WORDS, dtype=torch.int64, device=torch.device("cuda", d))
            st["next"] = 0
        t = blk[st["next"]:st["next"] + _PIN_WORDS]
        st["next"] += _PIN_WORDS
        return (t if dtype == torch.int64 else t.view(dtype))[:n]
    if _pin["buf"] is None:
        _pin["buf"] = torch.zeros((_PIN_SLOTS, _PIN_WORDS), dtype=torch.int64).pin_memory()
    slot = _pin["buf"][_pin["next"]]
    _pin["next"] = (_pin["next"] + 1) % _PIN_SLOTS
    slot.zero_()
    t = slot if dtype == torch.int64 else slot.view(dtype)
    return t[:n]


def cptr(t):
    """Pointer of a pinned counter as a kernel argument."""
    if not (t.is_pinned() or t.is_cuda):
        raise PccError("counter() tensors only")
    return t.data_ptr()


def read(t):
    """Values of a counter once the kernels that write it have finished."""
    if t.is_cuda:
        return t.tolist()
    torch.cuda.current_stream().synchronize()
    return t.tolist()


def read_many(counters):
    """Values of several counter() tensors with ONE device->host copy (they are cut from one zeroed block, so the span
    between the first and the last is a single contiguous read).  Returns a list of lists, one per counter."""
    if not counters:
        return []
    if all(t.is_cuda and t.dtype == torch.int64 for t in counters):
        base = counters[0].untyped_storage().data_ptr()
        if all(t.untyped_storage().data_ptr() == base for t in counters):
            lo = min(t.storage_offset() for t in counters)
            hi = max(t.storage_offset() + t.numel() for t in counters)
            if hi - lo <= 4096:
                blk = torch.empty(0, dtype=torch.int64, device=counters[0].device).set_(counters[0].untyped_storage(), lo, (hi - lo,))
                v = blk.tolist()
                return [v[t.storage_offset() - lo:t.storage_offset() - lo + t.numel()] for t in counters]
    return [read(t) for t in counters]


# ---- arithmetic form of the matrix products (include/pcc_hip.h PCC_ARITH_*) ----------------------------
# The library keeps no arithmetic state: every convolution call names its form and its guard word.  On the host side the form
# is a property of the calling THREAD's current scope (`arith_scope`), read by the wrappers in sparse.py at every call.
ARITH_F32, ARITH_BF6, ARITH_H3 = 0, 1, 2
H_GUARD_BUDGET = 2.5e-5     # PCC_H_GUARD_BUDGET: absolute error the scales of a tile may admit before the call is repeated in the
#                             six-term form: a quarter of the 1e-4 parity bar (cin * 2^-27 * max|row| * max|column|: 26 for cin = 128;
#                             the benchmark's layers stay below 0.5, tools/fp16_ranges.py)
ARITH_DEFAULT = {"f32": ARITH_F32, "bf6": ARITH_BF6, "h3": ARITH_H3}[os.environ.get("PCC_ARITH", "h3")]
# Diagnostic override (bench.py's `ms_per_step_strict` / `ms_per_step_fp32_mfma`, tests that compare forms): when not None EVERY
# call of the process takes this form, pinned scopes included -- encoder and decoder alike, so the two stay in step.
ARITH_FORCE = None
_tls = threading.local()


class RangeGuardTripped(Exception):
    """An fp16-pair product saw operands whose range admits more than H_GUARD_BUDGET of absolute error."""


def h_guard(device):
    """The guard word of `device` for the calling thread (int32 [1], zero unless a launch of an H3 scope tripped it)."""
    words = getattr(_tls, "guards", None)
    if words is None:
        words = _tls.guards = {}
    g = words.get(device)
    if g is None:
        g = words[device] = torch.zeros(1, dtype=torch.int32, device=device)
        words[device, "ptr"] = g.data_ptr()
    return g


class arith_scope:
    """`with arith_scope(ARITH_BF6): ...` -- the matrix products queued by this thread inside the block take that form.
    `pinned=True` marks a scope whose form is part of a CONTRACT (the hyper-synthesis: encoder and decoder must produce the
    same bits), which enclosing / fallback scopes must not change; only the diagnostic ARITH_FORCE overrides it."""

    def __init__(self, form, pinned=False):
        self.form, self.pinned = form, pinned

    def __enter__(self):
        self.prev = getattr(_tls, "scope", None)
        if self.prev is None or not self.prev[1]:          # inside a pinned scope nothing changes
            _tls.scope = (self.form, self.pinned)
        return self

    def __exit__(self, *exc):
        _tls.scope = self.prev
        return False


def arith():
    """Form of the calling thread's current scope."""
    if ARITH_FORCE is not None:
        return ARITH_FORCE
    sc = getattr(_tls, "scope", None)
    return ARITH_DEFAULT if sc is None else sc[0]


def arith_args(device):
    """(arith, d_guard) -- the two arguments every convolution entry point takes before its stream."""
    a = arith()
    if a != ARITH_H3:
        return a, None
    words = getattr(_tls, "guards", None)
    if words is None or (device, "ptr") not in words:
        h_guard(device)
        words = _tls.guards
    return a, words[device, "ptr"]


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)


def device_info():
    lib = load()
    cu = C.c_int(0)
    arch = C.create_string_buffer(64)
    check(lib.pcc_device_info(C.byref(cu), arch, 64), "pcc_device_info")
    return cu.value, arch.value.decode()
