#!/usr/bin/env python3
"""The training step of BASELINE configs[3] alone (bench.train_step_ms), for rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
r = bench.train_step_ms(torch.device("cuda:0"), steps=steps, warmup=6)
print(r)
