#!/usr/bin/env python3
"""Host time of the training step by autograd node and by operator (torch profiler, CPU side), one step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
one, info = bench.train_step_setup(dev)
for _ in range(5):
    one()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        one()
    torch.cuda.synchronize()
ka = prof.key_averages()
print("== autograd nodes (CPU total, 3 steps) ==")
rows = [e for e in ka if e.key.startswith("autograd::engine::evaluate_function")]
rows.sort(key=lambda e: -e.cpu_time_total)
tot = 0
for e in rows[:30]:
    print(f"{e.cpu_time_total / 3e3:8.3f} ms/step  x{e.count / 3:5.1f}  {e.key[37:]}")
for e in rows:
    tot += e.cpu_time_total
print(f"all nodes: {tot / 3e3:.2f} ms/step")
print("== operators by self CPU ==")
rows = sorted(ka, key=lambda e: -e.self_cpu_time_total)
for e in rows[:40]:
    print(f"{e.self_cpu_time_total / 3e3:8.3f} ms/step  x{e.count / 3:6.1f}  {e.key[:80]}")
