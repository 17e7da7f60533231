#!/usr/bin/env python3
"""Which Python lines of the training step make the host wait or copy?  torch profiler with stacks: host<->device copies,
nonzero (a device->host read of the count), item / tolist."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    one()
    torch.cuda.synchronize()
names = ("aten::_to_copy", "aten::nonzero", "aten::_local_scalar_dense", "aten::index_put_", "aten::_index_put_impl_", "aten::index")
rows = [e for e in prof.key_averages(group_by_stack_n=12) if e.key in names]
rows.sort(key=lambda e: -e.cpu_time_total)
for e in rows[:40]:
    st = [s for s in e.stack if "unified_point_cloud" in s or "bench.py" in s or "loss.py" in s][:4]
    print(f"{e.key:28s} x{e.count:<3d} cpu {e.cpu_time_total / 1e3:7.3f} ms  dev {e.device_time_total / 1e3:7.3f} ms")
    for s in st:
        print("      ", s.split("unified_point_cloud_compression_amd/")[-1])
