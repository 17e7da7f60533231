#!/usr/bin/env python3
"""Where does the host spend the training step?  cProfile over the step of bench.train_step_setup (model construction excluded)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
one, info = bench.train_step_setup(dev)
for _ in range(6):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    one()
torch.cuda.synchronize()
print(f"step {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms", info)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    one()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
print("---- by own time (5 steps) ----")
st.sort_stats("tottime").print_stats(45)
print("---- by cumulative time (5 steps) ----")
st.sort_stats("cumulative").print_stats(60)
