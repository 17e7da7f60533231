#!/usr/bin/env python3
"""Host-side timeline of one step: when each library call / host read / torch allocation is issued (perf_counter), so that the
host-bound phases (start of compress / decompress, after every size read) can be read off call by call.
usage: python tools/host_timeline.py [out.txt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unified_point_cloud_compression_amd import synth, lib as L

dev = torch.device("cuda:0")
model = bench.build_model(dev)
pc = torch.from_numpy(synth.surface_cloud(0, 10)).to(dev)
q = torch.tensor([[0.5, 0.5]], device=dev)
for _ in range(4):
    bench.step(model, pc, q)
torch.cuda.synchronize()
log = []
T = time.perf_counter
orig_call = L.call
def call(name, *a):
    t0 = T(); r = orig_call(name, *a); log.append((t0, T(), "call " + name)); return r
L.call = call
import unified_point_cloud_compression_amd.sparse as S
for mod in list(sys.modules.values()):
    if mod and getattr(mod, "__name__", "").startswith("unified_point_cloud_compression_amd") and getattr(mod, "L", None) is L:
        pass
orig_tolist = torch.Tensor.tolist
def tolist(self):
    t0 = T(); r = orig_tolist(self); log.append((t0, T(), "READ tolist")); return r
torch.Tensor.tolist = tolist
orig_cpu = torch.Tensor.cpu
def cpu(self, *a, **k):
    t0 = T(); r = orig_cpu(self, *a, **k); log.append((t0, T(), "READ cpu")); return r
torch.Tensor.cpu = cpu
orig_item = torch.Tensor.item
def item(self):
    t0 = T(); r = orig_item(self); log.append((t0, T(), "READ item")); return r
torch.Tensor.item = item
t_begin = T()
out = model.compress(pc, q)
log.append((T(), T(), "==== compress returned"))
dec = model.decompress(coordinates=bench.plain(out[3]), strings=out[0], shape=out[1], k=out[2], q_vals=out[4])
log.append((T(), T(), "==== decompress returned"))
torch.cuda.synchronize()
t_end = T()
lines = [f"step wall {1e3 * (t_end - t_begin):.2f} ms; {sum(1 for e in log if e[2].startswith('call'))} library calls, "
         f"{sum(1 for e in log if e[2].startswith('READ'))} reads"]
prev = t_begin
for t0, t1, name in log:
    lines.append(f"{1e6 * (t0 - t_begin):9.1f} us  +{1e6 * (t0 - prev):7.1f} since previous  took {1e6 * (t1 - t0):8.1f}  {name}")
    prev = t1
open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/host_timeline.txt", "w").write("\n".join(lines) + "\n")
print(lines[0])
