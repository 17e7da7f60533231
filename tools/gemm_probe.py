#!/usr/bin/env python3
"""Plain GEMM through the MFMA conv kernel (identity map): gemm_probe.py M K N -> ms, TFLOP/s (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unified_point_cloud_compression_amd import sparse as S
M, K, N = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev)
w = torch.nn.Parameter(torch.randn(K, N, device=dev) * 0.05)
pk = S.PackedConv().get(w)
for _ in range(2):
    y = S.conv_forward(x, pk, None, 1, K, N, None, M)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    y = S.conv_forward(x, pk, None, 1, K, N, None, M)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"M {M} K {K} N {N}: {ms:.3f} ms  {2e-9 * M * K * N / ms:.1f} TFLOP/s  out {M * N * 4 / 1e9:.2f} GB -> {M * N * 4 / ms / 1e6:.0f} GB/s written")
