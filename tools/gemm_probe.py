#!/usr/bin/env python3
"""Dense GEMM [n, cin] x [cin, ncol] through the library's MFMA path (K = 1 convolution), for profiling the generative
transposed convolutions' GEMM half: gemm_probe.py N CIN NCOL [reps].  PCC_DBG selects diagnostic variants."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unified_point_cloud_compression_amd import sparse as S  # noqa: E402

n, cin, ncol = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
x = torch.randn(n, cin, device=dev)
w = torch.nn.Parameter(torch.randn(cin, ncol, device=dev) * 0.05)
pk = S.PackedConv().get(w)
S.conv_forward(x, pk, None, 1, cin, ncol, None, n)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    S.conv_forward(x, pk, None, 1, cin, ncol, None, n)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"PCC_DBG={os.environ.get('PCC_DBG', '0')} arith={os.environ.get('PCC_ARITH', 'h3')} n {n} cin {cin} ncol {ncol}: {ms:.3f} ms  "
      f"{2.0 * n * cin * ncol / ms / 1e9:.1f} TFLOP/s  out {n * ncol * 4 / ms / 1e6:.0f} GB/s")
if os.environ.get("PCC_DBG", "0") == "0":
    got = S.conv_forward(x, pk, None, 1, cin, ncol, None, n)
    rows = torch.arange(0, n, max(n // 2048, 1), device=dev)
    want = (x[rows].double() @ w.detach().double())
    err = (got[rows].double() - want).abs()
    print(f"   check on {len(rows)} rows: max abs err {err.max().item():.3e}  (max |want| {want.abs().max().item():.2f})")
    if err.max().item() > 1e-3:
        bad = err > 1e-3
        br, bc = bad.any(1).nonzero().flatten(), bad.any(0).nonzero().flatten()
        print("   bad rows (sampled idx):", br[:20].tolist(), "... n", len(br), " bad cols:", bc[:40].tolist(), "... n", len(bc))
        print("   bad col blocks of 32:", sorted(set((bc // 32).tolist()))[:60])
        print("   bad row mod 128 (of sampled rows):", sorted(set((rows[br] % 128).tolist()))[:130])
