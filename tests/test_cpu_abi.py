"""CPU suite: the C-ABI library loads and exports every symbol include/pcc_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pcc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pcc_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from unified_point_cloud_compression_amd import lib
    names = _declared()
    assert len(names) >= 40
    so = ctypes.CDLL(lib.LIB_PATH)
    for nme in names:
        assert hasattr(so, nme), f"{nme} declared in pcc_hip.h but not exported by libpcc_hip.so"
    assert sorted(lib.SIGNATURES) == names, set(lib.SIGNATURES) ^ set(names)   # ctypes table mirrors the header 1:1
    lib.load()
    assert lib.load().pcc_version() >= 100


def test_size_queries_are_pure_host_functions():
    from unified_point_cloud_compression_amd import lib
    L = lib.load()
    assert L.pcc_sort_ws_bytes(1000) > 8000 and L.pcc_map_nbr_elems(1000, 5, 1, 0) == 125000
    assert L.pcc_map_nbr_elems(1000, 5, 2, 1) == 27000 and L.pcc_map_nbr_elems(1000, 2, 2, 1) == 1000
    # fp32 image + three bf16 planes + (pair-GEMM convolutions) two scaled fp16 planes + a scale per (offset, column)
    assert L.pcc_conv_packed_elems(125, 128, 128) == 125 * 128 * 128 * 5 // 2 + 125 * 128 * 128 + 125 * 128
    assert L.pcc_conv_packed_elems(27, 192, 192) == 27 * 192 * 256 * 5 // 2     # padded to the 128-wide column tile
    assert L.pcc_conv_packed_elems(125, 4, 128) == 125 * 4 * 128                # narrow inputs: fp32 image only
    assert L.pcc_conv_packed_elems(27, 24, 24) == 0                           # unsupported shape is reported, not guessed
    assert L.pcc_convt_packed_elems(125, 128, 32) == 128 * 4096 * 5 // 2 + 128 * 4096 + 4096   # + fp16 planes + column scales
    assert L.pcc_gdn_packed_elems(128) == 128 * 128 * 5 // 2 and L.pcc_gdn_packed_elems(24) == 0


def test_round4_training_queries_are_pure_host_functions():
    """Support predicates and workspace queries of the round-4 training entry points (no launch, no GPU needed)."""
    from unified_point_cloud_compression_amd import lib
    L = lib.load()
    # self-mapped weight gradient: odd K <= 27; one logit from 16 / 32 / 64 channels, 16 columns from 16 / 32
    assert L.pcc_conv_wgrad_self_supported(27, 16, 1) and L.pcc_conv_wgrad_self_supported(27, 64, 1)
    assert L.pcc_conv_wgrad_self_supported(27, 32, 16) and L.pcc_conv_wgrad_self_supported(1, 16, 16)
    assert not L.pcc_conv_wgrad_self_supported(27, 64, 16) and not L.pcc_conv_wgrad_self_supported(8, 32, 1)
    assert not L.pcc_conv_wgrad_self_supported(125, 32, 1) and not L.pcc_conv_wgrad_self_supported(27, 32, 3)
    # partial blocks: >= 1, bounded, and the query covers them (K * cin * cout floats per block)
    small, big = L.pcc_conv_wgrad_self_ws_bytes(100, 27, 16, 1), L.pcc_conv_wgrad_self_ws_bytes(10 ** 7, 27, 16, 1)
    assert 27 * 16 * 4 <= small < big <= 1024 * 27 * 16 * 4 + 256
    assert L.pcc_conv_wgrad_self_ws_bytes(10 ** 7, 27, 32, 16) <= 1024 * 27 * 32 * 16 * 4 + 256
    assert L.pcc_quant_mlp_params() == 2 * 10 + 10 + 10 * 10 + 10 + 10 + 1
    assert 151 * 4 <= L.pcc_quant_mlp_ws_bytes(1) < L.pcc_quant_mlp_ws_bytes(10 ** 8) <= 1024 * 151 * 4 + 256
    # top-k workspace: the per-segment state of one batch of launches + two (greater, equal) count arrays
    assert L.pcc_topk_ws_bytes(0) >= 48 * (256 + 2048 * 4)
    assert L.pcc_topk_ws_bytes(10 ** 7) - L.pcc_topk_ws_bytes(0) >= 2 * (10 ** 7 // 2048) * 8


def test_no_cpu_fallback():
    import torch
    from unified_point_cloud_compression_amd import lib
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    with pytest.raises(lib.PccError):
        lib.ptr(torch.zeros(3))
    with pytest.raises(lib.PccError):
        ME.SparseTensor(coordinates=torch.zeros((2, 4), dtype=torch.int32), features=torch.zeros((2, 1)))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "unified_point_cloud_compression_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(d, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(d, f)


GRID = (1, 3, 4, 8, 16, 32, 64, 128, 192, 256)


def _mfma_ok(cin, cout):
    return cout > 4 and (cin in (4, 8, 16) or (cin >= 32 and cin % 32 == 0))


def _cout_pad(cout):
    bn = 128 if cout >= 128 else (64 if cout > 32 else 32)
    return (cout + bn - 1) // bn * bn


def _layout_extent(K, cin, cout):
    """Floats the pack kernel of each kernel kind writes, restated from the documented layouts:
    thin [K][cout][cin]; wave16 [K][16][cin] (weights resident in LDS: K*16*(cin+4)*4 <= 64 KB); MFMA
    [K*cin/CB][cout_pad][CB]."""
    if cout <= 4:
        return K * cin * cout
    if cout <= 16 and cin in (16, 32, 64) and K * 16 * (cin + 4) * 4 <= 64 * 1024:
        return K * 16 * cin
    if not _mfma_ok(cin, cout):
        return 0
    total = _mfma_total(K * cin * _cout_pad(cout), cin)
    if K >= 64 and cin % 32 == 0 and cin <= 256 and cout % 4 == 0 and cout >= 128:
        total += K * cin * _cout_pad(cout) + K * _cout_pad(cout)       # pair-GEMM convolutions: + fp16 planes + (offset, column) scales
    return total


def _mfma_total(fp32_elems, cin):
    """MFMA weight image: fp32 layout, plus (cin a multiple of 32) three bf16 planes of the split path = 1.5x floats."""
    return fp32_elems + fp32_elems // 2 * 3 if cin % 32 == 0 else fp32_elems


def test_packed_size_queries_cover_what_the_pack_kernels_write():
    """Round 1 lost a GPU process to a packed buffer sized with the wrong query (GDN(16) sized by the conv query: 256
    floats for a 512-float MFMA layout).  For every kernel kind over the channel grid: the size query
    equals the layout's extent, and the pack entry points refuse (PCC_EWS, before any launch) a buffer one float short --
    so an undersized buffer can no longer be written past, whichever query the caller used."""
    from unified_point_cloud_compression_amd import lib
    L = lib.load()
    dummy = (ctypes.c_float * 4)()
    ptr = ctypes.cast(dummy, ctypes.c_void_p)
    EWS = -3
    checked = 0
    for K in (1, 8, 27, 125):
        for cin in GRID:
            for cout in GRID:
                want = _layout_extent(K, cin, cout)
                got = L.pcc_conv_packed_elems(K, cin, cout)
                assert got == want, (K, cin, cout, got, want)
                if got > 0:
                    assert L.pcc_conv_pack_weights(ptr, K, cin, cout, ptr, got - 1, None) == EWS
                    checked += 1
                # generative transpose, input-stationary: one flat [cin, K*cout] GEMM operand
                wt = _mfma_total(cin * _cout_pad(K * cout), cin) if _mfma_ok(cin, K * cout) else 0
                if wt and cin % 32 == 0 and cin <= 256:                     # dense products: + two scaled fp16 planes + 1/scale per column
                    wt += cin * _cout_pad(K * cout) + _cout_pad(K * cout)
                assert L.pcc_convt_packed_elems(K, cin, cout) == wt, (K, cin, cout)
                if wt > 0:
                    assert L.pcc_convt_pack_weights(ptr, K, cin, cout, ptr, wt - 1, None) == EWS
                # thin two-pass form: projection buffer t[K*cout][n_in]
                if cout <= 4 and cin in (4, 8, 16, 32, 64) and K * cout * cin * 4 <= 48 * 1024:
                    assert L.pcc_conv_ws_bytes(1000, K, cin, cout) >= K * cout * 1000 * 4
    for c in GRID:
        want = _mfma_total(c * _cout_pad(c), c) if _mfma_ok(c, c) else 0
        assert L.pcc_gdn_packed_elems(c) == want, c
        if want:
            assert L.pcc_gdn_pack(ptr, ptr, c, 1e-6, ptr, want - 1, ptr, None) == EWS
            # the round-1 failure: the conv query of the same shape can be smaller than the GDN layout -> now refused
            conv_q = L.pcc_conv_packed_elems(1, c, c)
            if conv_q < want:
                assert L.pcc_gdn_pack(ptr, ptr, c, 1e-6, ptr, conv_q, ptr, None) == EWS
    assert L.pcc_conv_packed_elems(1, 16, 16) < L.pcc_gdn_packed_elems(16)      # the exact round-1 case
    assert checked > 100
