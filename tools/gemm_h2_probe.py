#!/usr/bin/env python3
"""The dense products of a composite level alone (k_gemm_h2 through pcc_convt_fwd_csr, event-timed inside the library):
gemm_h2_probe.py [N=58051] [CIN=128] [K=343] [COUT=64] [reps=6].  Env switches (PCC_STAGGER, PCC_DBG, ...) apply."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from unified_point_cloud_compression_amd import lib as L, sparse as S  # noqa: E402

a = [int(v) for v in sys.argv[1:]]
n, cin, K, cout, reps = (a + [58051, 128, 343, 64, 6][len(a):])[:5]
dev = torch.device("cuda:0")
x = torch.randn(n, cin, device=dev)
W = torch.nn.Parameter(torch.randn(K, cin, cout, device=dev) * 0.05)
pk = S.PackedConv(True).get(W)
n_out = 1024                                                   # a token gather-sum: the GEMM is what is measured
first = torch.arange(0, n_out + 1, dtype=torch.int32, device=dev)
pair_ids = torch.arange(0, n * K, dtype=torch.int32, device=dev)
S.convt_forward_csr(x, pk, None, K, cin, cout, (first, pair_ids), n_out)
torch.cuda.synchronize()
L.call("pcc_prof_enable", 1)
for _ in range(reps):
    S.convt_forward_csr(x, pk, None, K, cin, cout, (first, pair_ids), n_out)
torch.cuda.synchronize()
f = bench.collect_forms(L)
L.call("pcc_prof_enable", 0)
for nme, v in f.items():
    if v["launches"]:
        ms = v["ms"] / v["launches"]
        print(f"{nme}: {ms:.3f} ms per launch  {v['flops'] / v['launches'] / ms / 1e9:.0f} TFLOP/s  {v['bytes'] / v['launches'] / ms / 1e6:.0f} GB/s "
              f"(PCC_STAGGER={os.environ.get('PCC_STAGGER', '0')})")
