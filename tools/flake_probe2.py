#!/usr/bin/env python3
"""Replays tests/test_gpu_autograd.py::test_conv_gradients (all seven cases, in order) N times in one process and reports every
failing case with the rows involved: the strided 192 -> 192 case failed twice inside suite runs and never alone."""
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_autograd as T  # noqa: E402

CASES = [(16, 32, 3, 1, "relu"), (32, 16, 3, 1, "leaky"), (128, 128, 5, 2, None), (32, 1, 3, 1, None), (8, 3, 1, 1, None), (4, 16, 5, 2, None),
         (192, 192, 3, 2, "leaky")]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for it in range(n):
    for c in CASES:
        try:
            T.test_conv_gradients(*c)
        except AssertionError as e:
            bad += 1
            print(f"iteration {it} case {c}: {str(e)[:300]}", flush=True)
print("failures:", bad, "of", n * len(CASES))
