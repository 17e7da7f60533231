#!/usr/bin/env python3
"""Is the data gradient of the strided 192 -> 192 convolution reproducible?  (tests/test_gpu_autograd.py::test_conv_gradients
[192-192-3-2-leaky] failed once in a full-suite run of round 3 and passed in four others.)  Repeats forward + backward in one
process with the library scratch and the allocator dirtied in between, and compares every result with the first bit for bit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import coords as co
from tests.util import dev, t, n, cloud_keys
import unified_point_cloud_compression_amd.MinkowskiEngine as ME
from unified_point_cloud_compression_amd import lib as L, sparse as S

cin = cout = 192
ks, stride = 3, 2
keys = cloud_keys(cin + cout + ks, 16, 0.2, 1, batch=2)
rng = np.random.default_rng(1)
f = rng.standard_normal((len(keys), cin)).astype(np.float32)
torch.manual_seed(0)
mod = ME.MinkowskiConvolution(cin, cout, kernel_size=ks, stride=stride, bias=True, dimension=3).to(dev())
go = t(np.random.default_rng(7).standard_normal((len(co.stride_keys(keys, stride)), cout)).astype(np.float32))
# CPU reference of the data gradient (the test's own: gather + matmul + index_add under torch autograd)
from oracle import codec
out_keys = co.stride_keys(keys, stride)
pairs = codec.kernel_map_pairs(keys, out_keys, ks, 1)
fr = torch.from_numpy(f).requires_grad_(True)
Wr = mod.kernel.detach().cpu().reshape(ks ** 3, cin, cout)
ref = torch.zeros((len(out_keys), cout)) + mod.bias.detach().cpu()
for k, (i, o) in enumerate(pairs):
    if len(i):
        ref = ref.index_add(0, torch.from_numpy(o.astype(np.int64)), fr[torch.from_numpy(i.astype(np.int64))] @ Wr[k])
ref = torch.nn.functional.leaky_relu(ref, 0.01)
ref.backward(go.cpu())
ref_grad = fr.grad.numpy()
first = None
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    x = ME.SparseTensor(coordinates=t(co.unpack_keys(keys)), features=t(f).requires_grad_(True))
    cs = x._cset
    out_set = cs.stride(stride)
    kmap = cs.kernel_map(out_set, ks)
    mod.kernel.grad = None
    out = mod._apply_conv(x, out_set, kmap, act=L.ACT_LEAKY)
    out.backward(go)
    res = (out.detach().clone(), x._F.grad.clone(), mod.kernel.grad.clone())
    torch.cuda.synchronize()
    err = np.abs(n(res[1]) - ref_grad)
    nbad = int((err > 1e-4 + 1e-4 * np.abs(ref_grad)).sum())
    if nbad:
        rows = np.unique(np.nonzero(err > 1e-3)[0])
        print(f"iteration {it}: data gradient differs from the CPU reference in {nbad} elements, max {err.max():.3e}, rows {rows[:12].tolist()}", flush=True)
        bad += 1
    if first is None:
        first = res
    else:
        for nme, a, b in zip(("forward", "data gradient", "weight gradient"), res, first):
            if not torch.equal(a, b):
                d = (a - b).abs()
                rows = (d.reshape(d.shape[0], -1).amax(1) > 0).nonzero().flatten().tolist()
                print(f"iteration {it}: {nme} differs from the first run: max {d.max().item():.3e}, rows {rows[:12]} ({len(rows)} rows)", flush=True)
                bad += 1
    # dirty the scratch buffers: another convolution with other shapes, a few allocations of odd sizes
    m = 3000 + 517 * (it % 7)
    xx = torch.randn(m, 128, device=dev())
    w = torch.nn.Parameter(torch.randn(128, 128 * (1 + it % 3), device=dev()))
    S.conv_forward(xx, S.PackedConv().get(w), None, 1, 128, w.shape[1], None, m)
    junk = [torch.full((1 << (10 + (it + j) % 9),), float("nan"), device=dev()) for j in range(4)]
    del junk
print("mismatching results:", bad)
