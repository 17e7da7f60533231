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
    assert L.pcc_conv_packed_elems(125, 128, 128) == 125 * 128 * 128
    assert L.pcc_conv_packed_elems(27, 192, 192) == 27 * 192 * 256            # padded to the 128-wide column tile
    assert L.pcc_conv_packed_elems(27, 24, 24) == 0                           # unsupported shape is reported, not guessed
    assert L.pcc_convt_packed_elems(125, 128, 32) == 128 * 4096
    assert L.pcc_gdn_packed_elems(128) == 128 * 128 and L.pcc_gdn_packed_elems(24) == 0


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
