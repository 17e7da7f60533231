import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench
from unified_point_cloud_compression_amd import sparse as S, synth
dev = torch.device("cuda:0")
model = bench.build_model(dev)
pc = torch.from_numpy(synth.surface_cloud(0, 10)).to(dev)
q = torch.tensor([[0.5, 0.5]], device=dev)
orig_g, orig_c, orig_p, orig_r = S.convt_forward_csr_grid, S.convt_forward_csr, S.conv_forward, S.convt_forward_rows
def rep(tag, feats, K, cin, cout):
    f = feats.abs()
    rm = f.amax(dim=1)
    print(f"{tag}: rows {feats.shape[0]} cin {cin} K {K} cout {cout}  row-max: max {rm.max().item():.3g} median {rm.median().item():.3g}  "
          f"frac elements < 2^-8 rowmax {(f < rm[:, None] * 2**-8).float().mean().item():.3f}", flush=True)
def spy_g(feats, packed_w, bias, K, cin, cout, csr, out_set, act, ex_bias, slope=0.01):
    rep("csr_grid", feats, K, cin, cout); return orig_g(feats, packed_w, bias, K, cin, cout, csr, out_set, act, ex_bias, slope)
def spy_c(feats, packed_w, bias, K, cin, cout, csr, n_out, act=0, slope=0.01, **kw):
    rep("csr", feats, K, cin, cout); return orig_c(feats, packed_w, bias, K, cin, cout, csr, n_out, act, slope, **kw)
def spy_p(feats, packed_w, bias, K, cin, cout, kmap, n_out, act=0, slope=0.01):
    if K >= 64: rep("conv", feats, K, cin, cout)
    return orig_p(feats, packed_w, bias, K, cin, cout, kmap, n_out, act, slope)
def spy_r(feats, packed_w, bias, K, cin, cout, csr, n_out, act=0, slope=0.01, **kw):
    rep("rows", feats, K, cin, cout); return orig_r(feats, packed_w, bias, K, cin, cout, csr, n_out, act, slope, **kw)
S.convt_forward_csr_grid, S.convt_forward_csr, S.conv_forward, S.convt_forward_rows = spy_g, spy_c, spy_p, spy_r
bench.step(model, pc, q)
for nme, m in model.named_modules():
    k = getattr(m, "kernel", None)
    if k is not None and k.dim() == 3 and k.shape[1] >= 32:
        a = k.detach().abs()
        print(nme, tuple(k.shape), "max |w|", a.max().item(), "col-max median", a.amax(dim=(0, 1)).median().item())
