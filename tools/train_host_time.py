#!/usr/bin/env python3
"""Where does the host spend the training step?  cProfile over bench.train_step_ms's loop body."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
print(bench.train_step_ms(dev, steps=5, warmup=3))
pr = cProfile.Profile()
pr.enable()
r = bench.train_step_ms(dev, steps=5, warmup=1)
pr.disable()
print(r)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
