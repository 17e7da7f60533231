#!/usr/bin/env python3
"""How often does a random weight draw of tests/test_gpu_autograd.py::test_conv_gradients[192-192-3-2-leaky] put a LeakyReLU
pre-activation within fp32 summation noise of 0?  (CPU only.)  fp32 vs fp64 accumulation of the same sums disagree on a sign in
about one draw of 40, and 8 of 40 draws have an output with |pre| < 1e-6: two fp32 implementations with different summation
orders (the HIP kernel and the CPU reference) therefore disagree on one derivative in a few percent of the draws -- each such
output is an error of 0.99 * |go| * |W| (2-4e-2) in all 192 channels of the 4-5 input rows it touches: the round-3 failure
signature exactly (763 / 958 bad entries = 4 / 5 rows x 192).  The test now seeds its weights and gives outputs within 1e-5
of the kink no upstream gradient."""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import coords as co, codec
from tests.util import cloud_keys
import unified_point_cloud_compression_amd.MinkowskiEngine as ME
cin=cout=192; ks=3; stride=2
keys = cloud_keys(cin+cout+ks, 16, 0.2, 1, batch=2)
rng = np.random.default_rng(1)
f = rng.standard_normal((len(keys), cin)).astype(np.float32)
out_keys = co.stride_keys(keys, stride)
pairs = codec.kernel_map_pairs(keys, out_keys, ks, 1)
flips=0; N=int(sys.argv[1])
for seed in range(N):
    torch.manual_seed(seed)
    mod = ME.MinkowskiConvolution(cin, cout, kernel_size=ks, stride=stride, bias=True, dimension=3)
    W = mod.kernel.detach().numpy().reshape(ks**3, cin, cout)*3
    b = mod.bias.detach().numpy().reshape(-1)
    o32 = np.zeros((len(out_keys), cout), np.float32) + b
    o64 = np.zeros((len(out_keys), cout), np.float64) + b
    for k,(i,o) in enumerate(pairs):
        if len(i):
            np.add.at(o32, o, f[i] @ W[k])
            np.add.at(o64, o, f[i].astype(np.float64) @ W[k].astype(np.float64))
    d = (o32>0) != (o64>0)
    m = np.abs(o64).min()
    print(seed, "sign flips f32 vs f64:", int(d.sum()), "min|pre|: %.2e"%m, "std %.2f"%o64.std(), "W absmax %.3f"%np.abs(W).max(), flush=True)
    flips += d.any()
print("seeds with a flip:", flips, "of", N)
