#!/usr/bin/env python3
"""Where the training step of BASELINE configs[3] spends its time, per LAYER: every library call of one step with its shapes
and its own duration (HIP events around the call, one step run call-synchronously), then torch's profiler table for the
operators that are not library calls (losses, optimiser, parametrisations).

usage: python3 tools/train_attribution.py [out.txt]
"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from unified_point_cloud_compression_amd import lib as L  # noqa: E402
from unified_point_cloud_compression_amd import sparse as S  # noqa: E402

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
dev = torch.device("cuda:0")
records = []
orig_call = L.call
WATCH = {"pcc_conv_wgrad": lambda a: f"n_in={a[1]} cin={a[2]} n_out={a[4]} cout={a[5]} K={a[6]}",
         "pcc_conv_fwd": None, "pcc_conv_fwd_pairs": None, "pcc_convt_fwd_csr": None, "pcc_convt_fwd": None,
         "pcc_gdn_fwd": None, "pcc_convt_scatter_rows": None}
state = {"on": False}


def timed_call(name, *args):
    if not state["on"]:
        return orig_call(name, *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig_call(name, *args)
    e1.record()
    e1.synchronize()
    ints = [a for a in args if isinstance(a, int) and 0 <= a < (1 << 31)]
    desc = WATCH[name](args) if WATCH.get(name) else " ".join(str(i) for i in ints[:8])
    records.append((name, desc, e0.elapsed_time(e1) * 1e3))
    return r


L.call = timed_call
S.L.call = timed_call

# the step of bench.train_step_ms, kept callable: run it through the bench with a hook on the last step
import unified_point_cloud_compression_amd.autograd as AG  # noqa: E402
AG.L.call = timed_call

one, info = bench.train_step_setup(dev)
for _ in range(5):
    one()
torch.cuda.synchronize()
import time  # noqa: E402
t0 = time.time()
for _ in range(5):
    one()
torch.cuda.synchronize()
print("untimed:", {"train_step_ms": (time.time() - t0) / 5 * 1e3, **info}, file=out)
state["on"] = True
one()
torch.cuda.synchronize()
state["on"] = False
agg = collections.OrderedDict()
for name, desc, us in records:
    k = (name, desc)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += us
tot = sum(v[1] for v in agg.values())
print(f"\nlibrary calls of ONE step (call-synchronous, so the sum {tot / 1e3:.2f} ms exceeds the pipelined step):", file=out)
for (name, desc), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{us:9.1f} us  x{n:<3d} {name:28s} {desc}", file=out)
by = collections.Counter()
for (name, desc), (n, us) in agg.items():
    by[name] += us
print("\nby entry point:", file=out)
for name, us in by.most_common():
    print(f"{us / 1e3:8.3f} ms  {name}", file=out)

from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    one()
    one()
    torch.cuda.synchronize()
print("\ntorch profiler, 2 steps (sorted by device time):", file=out)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70), file=out)
