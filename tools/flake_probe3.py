#!/usr/bin/env python3
"""One fresh process = the gradient cases of tests/test_gpu_autograd.py once, in order (the flake shows only on a process's
first execution).  Prints one line; run it many times under different switches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_autograd as T  # noqa: E402

CASES = [(16, 32, 3, 1, "relu"), (32, 16, 3, 1, "leaky"), (128, 128, 5, 2, None), (32, 1, 3, 1, None), (8, 3, 1, 1, None), (4, 16, 5, 2, None),
         (192, 192, 3, 2, "leaky")]
if os.environ.get("ONLY_LAST"):
    CASES = CASES[-1:]
if os.environ.get("NO_GRID"):
    from unified_point_cloud_compression_amd import sparse as S
    S.USE_GRID = False
out = "ok"
for c in CASES:
    try:
        T.test_conv_gradients(*c)
    except AssertionError as e:
        out = f"FAIL {c}: {str(e)[:220]}"
        break
print(out, flush=True)
