#!/usr/bin/env python3
"""Why is the GPU idle during the training step?  One step under torch's profiler (CPU + device activities), its timeline
exported and read back: GPU-idle periods of >= 20 us are attributed to what the host was doing at their midpoint (innermost
CPU operator on the thread that launched the next kernel)."""
import collections
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
one, info = bench.train_step_setup(dev)
for _ in range(6):
    one()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    one()
    one()
    torch.cuda.synchronize()
path = os.path.join(tempfile.gettempdir(), "train_trace.json")
prof.export_chrome_trace(path)
ev = json.load(open(path))["traceEvents"]
gpu = sorted((e["ts"], e["ts"] + e["dur"], e["name"]) for e in ev if e.get("ph") == "X" and e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset"))
cpu = [(e["ts"], e["ts"] + e["dur"], e["name"], e.get("tid")) for e in ev if e.get("ph") == "X" and e.get("cat") in ("cpu_op", "user_annotation", "cuda_runtime", "python_function")]
t0, t1 = gpu[0][0], gpu[-1][1]
busy = 0.0
end = gpu[0][0]
gaps = []
prev = None
for s, e, n in gpu:
    if s > end:
        gaps.append((end, s, prev, n))
        busy += e - s
    else:
        busy += max(0.0, e - max(s, end))
    if e > end:
        end, prev = e, n
print(f"2 steps: span {(t1 - t0) / 2e3:.2f} ms/step, GPU busy {busy / 2e3:.2f} ms/step, idle {(t1 - t0 - busy) / 2e3:.2f} ms/step in {len(gaps) / 2:.0f} gaps/step")
by = collections.Counter()
cnt = collections.Counter()
for a, b, p, n in gaps:
    if b - a < 20:
        by["(gaps under 20 us)"] += b - a
        cnt["(gaps under 20 us)"] += 1
        continue
    mid = (a + b) / 2
    act = [c for c in cpu if c[0] <= mid <= c[1]]
    act.sort(key=lambda c: c[1] - c[0])              # innermost first
    names = [c[2] for c in act if not c[2].startswith("hip")][:1] or [c[2] for c in act][:1] or ["(no host op: python glue)"]
    outer = [c[2] for c in act if c[2].startswith(("autograd::engine", "Optimizer", "SparseConvFn", "GdnFn"))][-1:]
    key = names[0][:60] + ("  <  " + outer[0][:60] if outer and outer[0] != names[0] else "")
    by[key] += b - a
    cnt[key] += 1
for k, v in by.most_common(30):
    print(f"{v / 2e3:7.3f} ms/step  x{cnt[k] / 2:5.1f}  {k}")
