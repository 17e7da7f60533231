#!/usr/bin/env python3
"""Headline benchmark: points/sec encode+decode of one vox10 frame per GPU (BASELINE.json config 2).

A step = UnifiedModel.compress + UnifiedModel.decompress of one synthetic longdress-like 10-bit frame
(~0.79 M voxels, one block, R2 architecture `configs/CVPR_inverse_scaling_fixed_R2.yaml`, q = [[0.5, 0.5]]), input
resident in HBM, timed like `utils.py:454-465` (sync, wall clock, sync).  Each rank codes its own frame (weak
scaling, no data-path collective); rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, dense fp32-input matrix peak (= the vector rate)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md, dense bf16 MFMA peak (~2.5 PF)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md, HBM3E ~8 TB/s
SPLIT_TERMS = 6                 # bf16 MFMA terms per fp32 product on the split path (k_conv_mfma_bf)
BITS = 10

R2_CONFIG = {
    "entropy_model": dict(C_bottleneck=128, C_hyper_bottleneck=192, quantization_mode="ste", inverse_rescaling=False,
                          quantization_offset=False, entropy_bottleneck_vbr=False, adaptive_BN=False),
    "g_a": dict(C_in=4, N1=128, N2=128, N3=128, N4=128),
    "g_s": dict(C_out=3, N1=128, N2=128, N3=128, N4=128),
}


def init_model(seed=0, gain=3.0, coder="pcc_streams"):
    """The benchmark's model on the CPU: R2 architecture, torch-seeded default init, conv kernels scaled by `gain`
    (tests/golden/make_fullsize.py hands exactly these parameters to the oracle)."""
    import copy
    from unified_point_cloud_compression_amd.model import UnifiedModel
    from unified_point_cloud_compression_amd.MinkowskiEngine.modules import _ConvBase
    torch.manual_seed(seed)
    cfg = copy.deepcopy(R2_CONFIG)
    cfg["entropy_model"]["entropy_coder"] = coder
    model = UnifiedModel(cfg)
    with torch.no_grad():   # no trained weights ship (README.md:122): seeded random init, scaled so latents are not all zero
        for m in model.modules():
            if isinstance(m, _ConvBase):
                m.kernel.mul_(gain)
    return model


def build_model(device, seed=0, gain=3.0, coder="pcc_streams"):
    model = init_model(seed, gain, coder).to(device).eval()
    model.update()
    return model


def plain(coords):
    """What a decoder receives (`utils.py:461-465` hands `decompress` plain tensors): copies of the latent coordinates
    WITHOUT the encoder's coordinate set / cached maps that `y.C` carries as Python attributes."""
    return [c.clone() for c in coords]


def step(model, pc, q, reuse_encoder_sets=False):
    out = model.compress(pc, q, block_size=1024)
    coords = out[3] if reuse_encoder_sets else plain(out[3])
    rec = model.decompress(coordinates=coords, strings=out[0], shape=out[1], k=out[2], q_vals=out[4])
    return out, rec


def csrc_sha():
    """SHA-256 over the kernel sources (csrc/*.hip, *.h, include/pcc_hip.h): identifies the code a PMC measurement
    belongs to (`.git` does not travel to the GPU box, so a content hash stands in for the commit)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    pk = os.path.join(ROOT, "unified_point_cloud_compression_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(pk, "*.hip")) + glob.glob(os.path.join(pk, "*.h")) +
                    [os.path.join(ROOT, "include", "pcc_hip.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


PMC_FILE = os.path.join("profiles", "pmc_traffic.json")


def pmc_traffic(kernel="k_gemm_h2"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/pmc_traffic.py; FETCH_SIZE and
    WRITE_SIZE in separate passes, KiB units, gfx950 FETCH x2 correction), with the hash of the kernel sources it was
    collected on.  (None, reason) when no measurement is committed or when it belongs to other kernel code."""
    try:
        rec = json.load(open(os.path.join(ROOT, PMC_FILE)))
    except Exception:
        return None, "no PMC measurement committed"
    now = csrc_sha()
    if rec.get("csrc_sha") != now:
        return None, f"stale: measured on csrc {rec.get('csrc_sha')}, running csrc {now} (re-run tools/pmc_traffic.py)"
    k = (rec.get("by_kernel") or {}).get(kernel)
    if k is None:
        return None, f"{PMC_FILE} holds no entry for {kernel}"
    return float(k["read"] + k["write"]), (f"rocprofv3 PMC passes of `{rec.get('command')}` on csrc {now}, {PMC_FILE}: "
                                            f"{k['read'] / 1e9:.3f} GB read + {k['write'] / 1e9:.3f} GB written per launch of {kernel}")


def pmc_total(kernel):
    """(HBM bytes of ALL launches of `kernel` in the PMC run, launches) or (None, 0) -- for sums over the kernels of a unit."""
    try:
        rec = json.load(open(os.path.join(ROOT, PMC_FILE)))
    except Exception:
        return None, 0
    k = (rec.get("by_kernel") or {}).get(kernel)
    if rec.get("csrc_sha") != csrc_sha() or k is None:
        return None, 0
    return float(k["read"] + k["write"]) * k["launches"], int(k["launches"])


def mfma_shape(cin, cout):
    return cout > 4 and (cin in (4, 8, 16) or (cin >= 32 and cin % 32 == 0))


def account_flops(model, pc, q):
    """One un-timed step with pair counting on: algorithmic FLOPs (2*P*Cin*Cout) of the MFMA conv launches."""
    from unified_point_cloud_compression_amd import sparse as S
    calls = []
    orig, orig_t, orig_c = S.conv_forward, S.convt_forward, S.convt_forward_csr

    def spy(feats, packed_w, bias, K, cin, cout, kmap, n_out, act=0, slope=0.01):
        calls.append((kmap, K, cin, cout, n_out, feats.shape[0]))
        return orig(feats, packed_w, bias, K, cin, cout, kmap, n_out, act, slope)

    def spy_t(feats, packed_w, bias, K, cin, cout, kmap, n_out, act=0, slope=0.01):
        calls.append((kmap, -K, cin, cout, n_out, feats.shape[0]))    # pairs = n_in*K: the dense GEMM does exactly the algorithmic FLOPs
        return orig_t(feats, packed_w, bias, K, cin, cout, kmap, n_out, act, slope)

    def spy_c(feats, packed_w, bias, K, cin, cout, csr, n_out, act=0, slope=0.01, **kw):
        calls.append((int(csr[0][n_out].item()), -K, cin, cout, n_out, feats.shape[0]))  # pairs of the map (full expansion: n_in*K)
        return orig_c(feats, packed_w, bias, K, cin, cout, csr, n_out, act, slope, **kw)

    orig_r, orig_h = S.convt_forward_rows, S.conv_head_forward

    def spy_r(feats, packed_w, bias, K, cin, cout, csr, n_out, act=0, slope=0.01, **kw):
        calls.append((int(csr[0][n_out].item()), float(K), cin, cout, n_out, feats.shape[0]))
        return orig_r(feats, packed_w, bias, K, cin, cout, csr, n_out, act, slope, **kw)

    def spy_h(feats, packed_w0, bias0, cmid, w2, bias2, cset, kmap):    # fused head: the cin -> cmid convolution is the MFMA launch
        calls.append((kmap, 27, feats.shape[1], cmid, feats.shape[0], feats.shape[0]))
        return orig_h(feats, packed_w0, bias0, cmid, w2, bias2, cset, kmap)

    orig_g = S.convt_forward_csr_grid

    def spy_g(feats, packed_w, bias, K, cin, cout, csr, out_set, act, ex_bias, slope=0.01):
        calls.append((S.csr_pair_total(csr, out_set.n), -K, cin, cout, out_set.n, feats.shape[0]))
        return orig_g(feats, packed_w, bias, K, cin, cout, csr, out_set, act, ex_bias, slope)

    names = ("conv_forward", "convt_forward", "convt_forward_csr", "convt_forward_rows", "conv_head_forward",
             "convt_forward_csr_grid")
    from unified_point_cloud_compression_amd import lib as L
    import ctypes as C
    S.COUNT_PAIRS = True
    for nme, f in zip(names, (spy, spy_t, spy_c, spy_r, spy_h, spy_g)):
        setattr(S, nme, f)
    L.call("pcc_prof_enable", 1)
    try:
        step(model, pc, q)
        torch.cuda.synchronize()
        seq = (C.c_int32 * 4096)()
        nseq = L.load().pcc_prof_sequence(seq, 4096)
        forms = [L.FORM_NAMES[seq[i]] for i in range(min(nseq, 4096))]
        forms = [f for f in forms if f != "k_convt_gather_csr"]        # (timed, but not an MFMA launch: accounted with its unit below)
    finally:
        L.call("pcc_prof_enable", 0)
        S.COUNT_PAIRS = False
        for nme, f in zip(names, (orig, orig_t, orig_c, orig_r, orig_h, orig_g)):
            setattr(S, nme, f)
    # Which kernel form each launch took -- and with it the 16-bit MFMA FLOPs executed per algorithmic FLOP: 3 (scaled fp16
    # pairs: k_gemm_h2 / k_pair_h2), 6 (bf16 split), 2500 / 157.3 for the fp32-input kernels priced at their own peak -- comes
    # from the library (`pcc_prof_collect_forms`, recorded at the launch), not from a copy of its dispatch rules.
    flops, launches, pairs_total, alg_bytes, exec_flops = 0.0, 0, 0, 0.0, 0.0
    by_form = {}
    timed = [c for c in calls if mfma_shape(c[2], c[3])]
    if len(forms) != len(timed):        # the spies and the library's timed launches are 1:1 by construction; say so if not
        print(f"bench: {len(timed)} accounted launches but {len(forms)} timed by the library: forms unattributed", file=sys.stderr)
        forms = ["other"] * len(timed)
    for (kmap, K, cin, cout, n_out, n_in), form in zip(timed, forms):
        K = abs(int(K))
        p = kmap if isinstance(kmap, int) else (kmap.pairs() if kmap is not None else n_out)
        fl = 2.0 * p * cin * cout
        flops += fl
        exec_flops += fl * FORM_TERMS[form]
        # compulsory traffic of the layer: every feature row, weight, map entry, coordinate touched once
        by8d = 4.0 * (n_in * cin + n_out * cout + K * cin * cout) + 8.0 * p + 16.0 * (n_in + n_out)
        f = by_form.setdefault(form, {"flops": 0.0, "launches": 0, "bytes_8d": 0.0, "layers": []})
        f["flops"] += fl
        f["launches"] += 1
        f["bytes_8d"] += by8d
        f["layers"].append({"n_in": int(n_in), "n_out": int(n_out), "K": K, "cin": int(cin), "cout": int(cout), "pairs": int(p),
                            "alg_gflop": fl / 1e9, "alg_gbytes_8d": by8d / 1e9})
        alg_bytes += by8d
        pairs_total += p
        launches += 1
    return flops, launches, pairs_total, alg_bytes, exec_flops, by_form


FORM_TERMS = {"k_gemm_h2": 3.0, "k_pair_h2": 3.0, "k_gemm_bf2": 6.0, "pair_bf": 6.0, "k_conv_mfma_bf": 6.0,
              "k_conv_mfma": PEAK_BF16_MFMA_TFLOPS / PEAK_FP32_MFMA_TFLOPS, "k_conv_wave16": PEAK_BF16_MFMA_TFLOPS / PEAK_FP32_MFMA_TFLOPS,
              "other": 6.0}


def collect_forms(lib):
    """Event-timed launches since `pcc_prof_enable(1)`, by the kernel form each one took."""
    import ctypes as C
    n = len(lib.FORM_NAMES)
    ms, la, fl, by = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)(), (C.c_double * n)()
    lib.check(lib.load().pcc_prof_collect_forms(ms, la, fl, by), "pcc_prof_collect_forms")
    return {nme: {"ms": ms[i], "launches": la[i], "flops": fl[i], "bytes": by[i]} for i, nme in enumerate(lib.FORM_NAMES)}


def cpu_baseline(threads=None):
    """The oracle (numpy restatement, 'port'; MinkowskiEngine cannot run here) timed on this host's cores with the
    harness of BASELINE.md section 3 on a bounded sample of the same workload: the same synthetic surface at vox9
    (198 k points, a quarter of the benchmark frame), same R2 architecture and weights recipe, encode + decode,
    1 warm-up run, median of 3.  The benchmark's own vox10 frame takes the oracle ~160 s (build container, 6 threads:
    `oracle_seconds` of tests/golden/full_config2_vox10.npz, quoted in `sample`) -- too long for the default run."""
    from oracle import codec
    from unified_point_cloud_compression_amd import synth
    threads = threads or min(os.cpu_count() or 1, 16)           # the GPU box's CPU share for one GPU
    torch.set_num_threads(threads)
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=threads)
    except Exception:
        limit = None
    P = codec.random_params(codec.R2_CONFIG, 0, gain=3.0)
    q = np.array([[0.5, 0.5]], dtype=np.float32)

    def once(pc):
        t0 = time.time()
        blocks = codec.compress(P, codec.R2_CONFIG, pc, q, threads=threads)
        codec.decompress(P, codec.R2_CONFIG, blocks, threads=threads)
        return time.time() - t0

    pc9 = synth.surface_cloud(0, 9)
    once(pc9)                                                   # warm-up (BLAS thread pools, page faults of the work buffers)
    runs = sorted(once(pc9) for _ in range(3))
    t9 = runs[1]
    if limit is not None:
        limit.restore_original_limits()
    vox10 = ""
    try:
        fx = np.load(os.path.join(ROOT, "tests", "golden", "full_config2_vox10.npz"))
        sec = float(np.sum(fx["oracle_seconds"]))
        vox10 = (f"; the benchmark's vox10 frame ({int(fx['n_points'])} points): {sec:.0f} s = {int(fx['n_points']) / sec:.0f} points/s "
                 f"on the build container (6 threads, tests/golden/make_fullsize.py)")
    except Exception:
        pass
    return {"value": pc9.shape[0] / t9, "unit": "points/s", "cores": threads, "kind": "port",
            "runs_s": [round(r, 2) for r in runs],
            "sample": f"oracle (numpy restatement; MinkowskiEngine unavailable) encode+decode of the same synthetic surface at vox9 "
                      f"({pc9.shape[0]} points), R2 architecture: 1 warm-up, median of 3 = {t9:.1f} s on {threads} threads{vox10}"}


def oracle_figures(bits):
    """Rate / distortion of the ORACLE on this exact frame and these weights (tests/golden/full_config2_vox10.npz,
    produced in the build container by tests/golden/make_fullsize.py): printed beside the build's own figures."""
    if bits != 10:
        return None
    try:
        fx = np.load(os.path.join(ROOT, "tests", "golden", "full_config2_vox10.npz"))
        n = int(fx["n_points"])
        return {"bpp_likelihood": (float(fx["bits_y"]) + float(fx["bits_z"])) / n, "d1_psnr_sym": float(fx["d1_sym"]),
                "d1_psnr_AB": float(fx["d1_AB"]), "d1_psnr_BA": float(fx["d1_BA"]), "n_points": n}
    except Exception:
        return None


def true_geometry_ms(model, pc, q, steps=5, warmup=2):
    """Auxiliary workload (never `value`): the same frame with the three occupancy levels driven by the GROUND-TRUTH
    geometry (mask = candidate in the down-sampled input), which is what a trained model's top-k converges to.  The
    generative sets then are ~1.2 M / 4.6 M rows instead of the 2.4 M / 14.5 M that random weights scatter, so this is
    the step time an `evaluate.py`-style run with trained weights would see from the sparse-convolution path."""
    from unified_point_cloud_compression_amd import lib as L
    x = model.block_input(pc)
    s1 = x._cset
    s2 = s1.stride(2)
    s4 = s2.stride(4)
    gt = [s4, s2, s1]
    sizes = {}

    def probe(stage, lvl, cset, logit, mask, feats):
        if stage != "select":
            return None
        g = gt[lvl]
        rows = torch.empty(max(cset.n, 1), dtype=torch.int32, device=pc.device)
        L.call("pcc_lookup_rows", L.ptr(g.keys), g.n, L.ptr(cset.keys), cset.n, L.ptr(rows), L.stream())
        sizes[lvl] = cset.n
        return rows[:cset.n] >= 0

    def one():
        out = model.compress(pc, q, block_size=1024)
        return model.decompress(coordinates=plain(out[3]), strings=out[0], shape=out[1], k=out[2], q_vals=out[4], probe=probe)

    for _ in range(warmup):
        rec = one()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        rec = one()
    torch.cuda.synchronize()
    ms = (time.time() - t0) / steps * 1e3
    exact = rec.shape[0] == pc.shape[0]
    return {"ms_per_step_true_geometry": ms, "candidate_rows": [sizes.get(i) for i in range(3)], "lossless_geometry": bool(exact)}


def train_step_setup(device, bottleneck_step=True, fused_adam=False):
    """The training step of BASELINE configs[3] as a callable: `one()` runs forward, losses, backward, gradient clipping and the
    model optimiser's step, then the quantile (aux) loss with the bottleneck optimiser's step, reading both losses as
    `train.py:196-236` does.  4 cubes of 128^3 cut from the benchmark frame, `configs/CVPR_inverse_scaling.yaml` (adaptive
    bottleneck, quantisation offsets, inverse rescaling, STE)."""
    import copy
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    from unified_point_cloud_compression_amd import synth
    from unified_point_cloud_compression_amd.loss import Loss
    from unified_point_cloud_compression_amd.model import UnifiedModel
    cfg = copy.deepcopy(R2_CONFIG)
    cfg["entropy_model"].update(adaptive_BN=True, quantization_offset=True, inverse_rescaling=True)
    loss_cfg = {"Multiscale_FocalLoss": {"type": "Multiscale_FocalLoss", "alpha": 0.5, "gamma": 2.0},    # `configs/CVPR_inverse_scaling.yaml:58-75`
                "ColorLoss": {"type": "ColorLoss", "loss": "L2"},
                "bpp-y": {"type": "BPPLoss", "key": "y", "weight": 1.0},
                "bpp-z": {"type": "BPPLoss", "key": "z", "weight": 1.0}}
    torch.manual_seed(0)
    model = UnifiedModel(cfg).to(device).train()
    pc = synth.surface_cloud(0, 10, shuffle=False)
    cubes = []
    for origin in ((512, 300, 500), (300, 512, 420), (640, 512, 600), (512, 512, 300)):
        o = np.array(origin)
        m = np.all((pc[:, :3] >= o) & (pc[:, :3] < o + 128), axis=1)
        if m.sum() >= 300:
            cubes.append(pc[m])
    coords, feats = ME.utils.sparse_collate([c[:, :3] - c[:, :3].min(0) for c in cubes], [c[:, 3:] for c in cubes])
    nb = len(cubes)
    q = torch.tensor([[0.4, 0.7]] * nb, device=device)
    Lam = torch.tensor([[5.0, 400.0]] * nb, device=device)
    # `train.py:96-112`: the model's parameters and the bottleneck's quantiles have an optimiser each
    kw = {"fused": True} if fused_adam else {}     # (reference: the default `optim.Adam(params, lr=...)`, `train.py:67-75`)
    opt = torch.optim.Adam([p for nme, p in model.named_parameters() if not nme.endswith(".quantiles")], lr=1e-4, **kw)
    opt_aux = torch.optim.Adam([p for nme, p in model.named_parameters() if nme.endswith(".quantiles")], lr=1e-3, **kw)
    loss_fn = Loss(copy.deepcopy(loss_cfg))
    coords, feats = coords.to(device), feats.float().to(device)

    def one():
        opt.zero_grad(set_to_none=True)
        opt_aux.zero_grad(set_to_none=True)
        x = ME.SparseTensor(coordinates=coords, features=feats)
        out = model(x, q, Lam)
        total, _ = loss_fn(x, out)
        value = total.item()                          # `train.py:221`: read before the backward pass
        total.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        if not bottleneck_step:                       # (what rounds 1-3 timed: the model optimiser's half of the step)
            return value, None
        aux = model.aux_loss()
        aux_value = aux.item()
        aux.backward()
        opt_aux.step()
        return value, aux_value

    info = {"cubes": nb, "points": int(coords.shape[0]),
            "config": "CVPR_inverse_scaling (adaptive_BN, offsets, inverse rescaling, STE), R2 width; model + bottleneck optimisers"}
    return one, info


def train_step_ms(device, steps=10, warmup=6):
    """Auxiliary (BASELINE configs[3], never `value`): wall time of `train_step_setup`'s step."""
    one, info = train_step_setup(device)
    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        last = one()
    torch.cuda.synchronize()
    out = {"train_step_ms": (time.time() - t0) / steps * 1e3, "loss": last[0], "aux_loss": last[1], **info}
    # the same step without the quantile loss / bottleneck optimiser, as rounds 1-3 reported it (comparison across rounds only)
    one2, _ = train_step_setup(device, bottleneck_step=False)
    for _ in range(warmup):
        one2()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        one2()
    torch.cuda.synchronize()
    out["train_step_ms_without_bottleneck_step"] = (time.time() - t0) / steps * 1e3
    # the full step with torch's single-kernel Adam (`fused=True`: same update rule, one launch per optimiser instead of ~25
    # foreach launches and a host->device copy of the step scalars) -- an implementation switch of torch, reported beside the
    # reference's default construction, never instead of it
    try:
        one3, _ = train_step_setup(device, fused_adam=True)
        for _ in range(warmup):
            one3()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(steps):
            one3()
        torch.cuda.synchronize()
        out["train_step_ms_fused_adam"] = (time.time() - t0) / steps * 1e3
    except (RuntimeError, TypeError) as e:            # (a torch build without the fused kernel)
        out["train_step_ms_fused_adam"] = None
        out["fused_adam_error"] = str(e)[:120]
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks through torch.distributed.run as a CHILD process,
    before this process touches the GPU (never exec / re-exec after HIP init), and leave with its exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bits", type=int, default=BITS)
    ap.add_argument("--coder", default="pcc_streams", choices=["pcc_streams", "ans", "symbols"],
                    help="entropy coder inside the timed region (default: per-channel GPU rANS)")
    ap.add_argument("--no-aux", action="store_true", help="skip the auxiliary (un-timed) true-geometry and train-step figures")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    from unified_point_cloud_compression_amd import frames as _frames
    cores = _frames.pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world))) if world > 1 else []
    # PCC_BENCH_REHEARSE=1: the N-rank control flow on ONE GPU -- every rank on device 0, collectives over gloo on host tensors.
    # A check that the ranks meet at the same collectives and that rank 0 prints its line; never a measurement (the line says so).
    rehearse = world > 1 and os.environ.get("PCC_BENCH_REHEARSE", "0") == "1"
    gpu_index = 0 if rehearse else local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", gpu_index))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(gpu_index)
    device = torch.device("cuda", gpu_index)
    coll_device = torch.device("cpu") if rehearse else device      # where the collectives' tensors live

    from unified_point_cloud_compression_amd import lib, synth, frames
    lib.load()
    cu, arch = lib.device_info()
    model = build_model(device, coder=args.coder)
    pc_np = synth.surface_cloud(seed=rank, bits=args.bits)
    pc = torch.from_numpy(pc_np).to(device)
    q = torch.tensor([[0.5, 0.5]], device=device)
    n_points = pc.shape[0]

    def barrier():
        if world > 1:
            dist.barrier() if rehearse else dist.barrier(device_ids=[gpu_index])

    for _ in range(args.warmup):
        step(model, pc, q)
    torch.cuda.synchronize()
    (flops_step, launches_step, pairs_step, alg_bytes_step, exec_flops_step, flops_by_form) = (
        account_flops(model, pc, q) if rank == 0 else (0.0, 0, 0, 0.0, 0.0, {}))

    def timed_loop(steps, reuse_encoder_sets=False, all_ranks=True):
        """EXACTLY `steps` steps between barrier + synchronize on both sides.  The decoder gets PLAIN coordinate tensors
        (`plain`): it rebuilds every coordinate set, map and pair list from them, as a decoder reading a bitstream must.
        all_ranks=False: a rank-0-only diagnostic pass -- no collective inside (the other ranks are not in this loop)."""
        if all_ranks:
            barrier()
        torch.cuda.synchronize()
        t0 = time.time()
        t_enc, marks = 0.0, []
        for _ in range(steps):
            te = time.time()
            out = model.compress(pc, q, block_size=1024)
            torch.cuda.synchronize()
            t_enc += time.time() - te
            coords = out[3] if reuse_encoder_sets else plain(out[3])
            rec = model.decompress(coordinates=coords, strings=out[0], shape=out[1], k=out[2], q_vals=out[4])
            marks.append(time.time())           # host time only (no sync): the next compress starts with a size read
        torch.cuda.synchronize()
        if all_ranks:
            barrier()
        return time.time() - t0, t_enc, [t0] + marks, out, rec

    # pass 1 -- the one `value` comes from: no event records, no profiler hooks inside
    lib.call("pcc_prof_enable", 0)
    dt, t_enc, step_marks, out, rec = timed_loop(args.steps)

    # every collective of the run happens HERE, right behind the timed pass: what follows is rank 0 alone (event pass, strict /
    # fp32 / cached passes, auxiliary figures, CPU baseline) and must not contain one -- the other ranks are already past them
    tt = torch.tensor([dt], dtype=torch.float64, device=coll_device)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt_max = float(tt.item())
    recs = frames.gather_records([(rank, n_points, t_enc / args.steps, (dt - t_enc) / args.steps, 0.0, rec.shape[0])],
                                 coll_device)
    # per-rank diagnostics for the first real multi-GPU run: where each rank's step went, on which cores it ran
    steps_ms = [(b - a) * 1e3 for a, b in zip(step_marks[:-1], step_marks[1:])]
    diag = frames.gather_records([(rank, t_enc / args.steps * 1e3, (dt - t_enc) / args.steps * 1e3, dt / args.steps * 1e3, min(steps_ms),
                                   max(steps_ms), len(cores), cores[0] if cores else -1)], coll_device, n_fields=8)
    total_points = sum(r[1] for r in recs)

    # pass 2 (rank 0, un-timed for `value`): the same steps with a HIP-event pair around every MFMA launch, recorded on the
    # launch stream inside the library -> launch durations by kernel form (roofline)
    forms = None
    if rank == 0:
        lib.call("pcc_prof_enable", 1)
        ev_steps = min(args.steps, 10)
        for _ in range(ev_steps):
            step(model, pc, q)
        torch.cuda.synchronize()
        forms = collect_forms(lib)
        lib.call("pcc_prof_enable", 0)
        for f in forms.values():
            f["ms_per_step"] = f.pop("ms") / ev_steps
            f["launches_per_step"] = f.pop("launches") / ev_steps
            f["flops"] /= ev_steps
            f["bytes"] /= ev_steps

    # pass 3 (rank 0): the same step under the stricter arithmetic switches, so that the figure behind `dtype: f32` stands
    # beside the headline: dense / pair products as six bf16 terms (24-bit split of both operands) instead of three fp16
    # terms; and every product on the fp32-input MFMA instructions
    strict_ms = fp32_ms = cached_dec_ms = None
    if rank == 0 and not args.no_aux:
        def quick(n=5, **kw):
            for _ in range(2):
                step(model, pc, q, **kw)
            d, te, _, _, _ = timed_loop(n, all_ranks=False, **kw)
            return d / n * 1e3, (d - te) / n * 1e3
        try:
            lib.ARITH_FORCE = lib.ARITH_BF6               # (process-wide diagnostic override: encoder and decoder alike)
            strict_ms, _ = quick()
            lib.ARITH_FORCE = lib.ARITH_F32
            fp32_ms, _ = quick()
        finally:
            lib.ARITH_FORCE = None
        _, cached_dec_ms = quick(reuse_encoder_sets=True)     # round-2 figure: decoder re-using the encoder's sets and maps

    # auxiliary (un-timed for `value`): the same step with integer symbols handed across the entropy-coder boundary,
    # i.e. the SURVEY 8a hot path alone (sparse convolutions + likelihood kernels), for comparison across rounds
    hot_ms = None
    if rank == 0 and args.coder != "symbols":
        m2 = build_model(device, coder="symbols")
        m2.load_state_dict(model.state_dict())
        for _ in range(2):
            step(m2, pc, q)
        torch.cuda.synchronize()
        t1 = time.time()
        for _ in range(3):
            step(m2, pc, q)
        torch.cuda.synchronize()
        hot_ms = (time.time() - t1) / 3 * 1e3
        del m2


    def count_bits(strings):   # `utils.count_bits` (utils.py:30-48)
        return sum(count_bits(x) if isinstance(x, list) else len(x) * 8 for x in strings)
    bpp = (count_bits(out[0]) / n_points) if args.coder != "symbols" else None

    # rate / distortion of the coded frame (un-timed): -sum log2 p / N (`loss.py:77-79`) and the D1 report of
    # `metrics/metric.py:113-118,74` on the GPU, next to the oracle's figures for the same frame and weights
    rd = None
    if rank == 0:
        from unified_point_cloud_compression_amd import metrics
        y, _ = model.g_a(model.block_input(pc))
        y_lik, z_lik = model.entropy_model.likelihoods(y, q)
        bits_lik = float(-torch.log2(y_lik.double()).sum().item() - torch.log2(z_lik.double()).sum().item())
        m = metrics.pointcloud_metrics(pc, rec, (1 << args.bits) - 1)
        rd = {"bpp_likelihood": bits_lik / n_points, "d1_psnr_sym": m["sym_psnr_mse"], "d1_psnr_AB": m["AB_psnr_mse"],
              "d1_psnr_BA": m["BA_psnr_mse"], "y_psnr_sym": m["sym_y_psnr"], "oracle": oracle_figures(args.bits)}

    if rank == 0:
        split_on = lib.ARITH_DEFAULT != lib.ARITH_F32
        ms_step = dt_max / args.steps * 1e3
        # ---- roofline (SURVEY 8d accounting) ----------------------------------------------------------------
        # Every event-timed kernel form is priced with the ALGORITHMIC work of the layers it ran: FLOP = 2 P Cin Cout and
        # BYTES = 4 (N_in Cin + N_out Cout + K Cin Cout) + 8 P + 16 (N_in + N_out) -- the layer's compulsory traffic; buffers
        # the design adds (the per-pair products T) count as `traffic`, never as useful bytes.  The DOMINANT kernel is the form
        # with the most time per step; `unit` prices the SURVEY 8d unit it belongs to (a composite level = dense products +
        # ordered gather-sum) the same way.
        kernels = {}
        for nme, f in forms.items():
            if f["launches_per_step"] <= 0 or nme == "k_convt_gather_csr":
                continue
            bf = flops_by_form.get(nme, {})
            fl, by8 = bf.get("flops", 0.0), bf.get("bytes_8d", 0.0)           # per step (one accounting step)
            sec = f["ms_per_step"] * 1e-3
            roof_t = PEAK_BF16_MFMA_TFLOPS / FORM_TERMS[nme]
            k = {"ms_per_step": f["ms_per_step"], "launches_per_step": f["launches_per_step"],
                 "alg_tflops": fl / sec / 1e12 if sec > 0 else None, "mfma_roof_tflops": roof_t,
                 "alg_gbs_8d": by8 / sec / 1e9 if sec > 0 else None}
            k["mfma_frac"] = k["alg_tflops"] / roof_t if k["alg_tflops"] else None
            k["hbm_frac"] = k["alg_gbs_8d"] / PEAK_HBM_GBS if k["alg_gbs_8d"] else None
            k["flop_per_byte"] = fl / by8 if by8 else None
            ridge = roof_t * 1e12 / (PEAK_HBM_GBS * 1e9)
            k["bound"] = "hbm" if (k["flop_per_byte"] is not None and k["flop_per_byte"] < ridge) else "mfma"
            if f["bytes"] > 0:      # dense products: operand + RESULT stream of the kernel itself (counts the product buffer T)
                k["kernel_stream_gbs"] = f["bytes"] / sec / 1e9
                k["kernel_stream_frac"] = k["kernel_stream_gbs"] / PEAK_HBM_GBS
            kernels[nme] = k
        dom = max(kernels, key=lambda n: kernels[n]["ms_per_step"]) if kernels else None
        d, df = (kernels[dom], forms[dom]) if dom else ({}, {})
        traffic, traffic_note = pmc_traffic(dom) if dom else (None, "no timed launches")
        conv_ms = sum(k["ms_per_step"] for k in kernels.values())
        fam_ach = flops_step / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else None
        fam_peak = PEAK_BF16_MFMA_TFLOPS * flops_step / exec_flops_step if exec_flops_step else PEAK_FP32_MFMA_TFLOPS
        lps = max(df.get("launches_per_step", 0), 1e-9)
        bf = flops_by_form.get(dom, {}) if dom else {}
        alg_b = bf.get("bytes_8d", 0.0) / lps if bf else None
        alg_f = bf.get("flops", 0.0) / lps if bf else None
        if d.get("bound") == "hbm":
            roof = {"bound": "hbm", "achieved": d.get("alg_gbs_8d"), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": d.get("hbm_frac")}
        else:
            roof = {"bound": "mfma", "achieved": d.get("alg_tflops"), "peak": d.get("mfma_roof_tflops"), "unit": "TFLOP/s",
                    "frac": d.get("mfma_frac")}
        # the SURVEY 8d unit of the composite levels: dense products (k_gemm_h2) + ordered gather-sum (k_convt_gather_csr)
        unit = None
        gat = forms.get("k_convt_gather_csr")
        if dom == "k_gemm_h2" and gat and gat["launches_per_step"] > 0:
            u_ms = df["ms_per_step"] + gat["ms_per_step"]
            # (the PMC run also holds the small gather-sums of the hyper-synthesis under the same kernel name: their traffic is
            #  negligible, so ALL gather-sum bytes of the run are divided by the run's composite levels = its k_gemm_h2 launches)
            g_bytes, _ = pmc_total("k_convt_gather_csr")
            m_bytes, m_launches = pmc_total("k_gemm_h2")
            u_note = "k_gemm_h2 + k_convt_gather_csr entries of " + PMC_FILE
            ut = ((m_bytes + g_bytes) / m_launches) if (g_bytes and m_bytes and m_launches) else None
            unit = {"name": "composite level = k_gemm_h2 + k_convt_gather_csr (pcc_convt_fwd_csr_grid)",
                    "units_per_step": lps, "ms_per_unit": u_ms / lps, "alg_flops_per_unit": alg_f, "alg_bytes_per_unit": alg_b,
                    "alg_tflops": bf["flops"] / (u_ms * 1e-3) / 1e12, "mfma_frac": bf["flops"] / (u_ms * 1e-3) / 1e12 / d["mfma_roof_tflops"],
                    "alg_gbs": bf["bytes_8d"] / (u_ms * 1e-3) / 1e9, "hbm_frac": bf["bytes_8d"] / (u_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                    "traffic": ut, "traffic_over_alg_bytes": (ut / alg_b) if (ut and alg_b) else None,
                    "traffic_source": u_note, "layers": bf.get("layers")}
        roof.update({
            "kernel": dom, "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_note,
            "alg_bytes_per_launch": alg_b, "alg_flops_per_launch": alg_f,
            "traffic_over_alg_bytes": (traffic / alg_b) if (traffic and alg_b) else None,
            "avg_launch_ms": df.get("ms_per_step", 0.0) / lps, "launches_per_step": df.get("launches_per_step"),
            "mfma_frac": d.get("mfma_frac"), "hbm_frac": d.get("hbm_frac"), "flop_per_byte": d.get("flop_per_byte"),
            "kernel_stream_frac": d.get("kernel_stream_frac"), "kernel_stream_gbs": d.get("kernel_stream_gbs"),
            "unit": unit,
            "note": ("dominant kernel = the event-timed form with the most time per step, priced by SURVEY 8d: algorithmic FLOPs = "
                     "2 x pairs x Cin x Cout of the layers it ran, algorithmic bytes = 4 (N_in Cin + N_out Cout + K Cin Cout) + 8 pairs + "
                     "16 (N_in + N_out) -- the per-pair product buffer T is NOT counted (it shows up in `traffic`); `kernel_stream_frac` "
                     "is the kernel's own operand + result stream (T included) against the HBM peak, the figure rounds 2-3 printed as "
                     "`frac`; `unit` = the composite level the kernel is half of.  Durations: HIP events on the launch stream, recorded "
                     "inside the library in a separate pass (never inside the loop `value` is timed on); mfma roof = dense 16-bit MFMA "
                     "peak 2500 TFLOP/s / MFMA terms per product of the form"),
            "kernels": kernels,
            # all event-timed MFMA launches together (round-2 style figure, kept for comparison across rounds)
            "mfma_family": {"achieved": fam_ach, "peak": fam_peak, "unit": "TFLOP/s", "frac": (fam_ach / fam_peak) if fam_ach else None,
                            "executed_16bit_tflops": (fam_ach * exec_flops_step / flops_step) if (fam_ach and flops_step) else None,
                            "vs_fp32_mfma_peak": (fam_ach / PEAK_FP32_MFMA_TFLOPS) if fam_ach else None,
                            "flop_per_step": flops_step, "pairs_per_step": pairs_step, "launches_per_step": launches_step,
                            "conv_ms_per_step": conv_ms, "alg_bytes_per_launch_survey_8d": (alg_bytes_step / launches_step) if launches_step else None},
        })
        line = {
            "metric": "points/sec encode+decode, longdress vox10, 1 GPU; bpp & D1-PSNR parity",
            "value": total_points * args.steps / dt_max,
            "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: one synthetic longdress-like vox{args.bits} frame per GPU "
                                   f"({n_points} points on rank 0), R2 architecture, random-init weights, q=[[0.5,0.5]], "
                                   f"1 block; entropy coder in the timed region: {args.coder}; the decoder starts from plain "
                                   f"coordinate tensors (no coordinate set, map or pair list of the encoder is re-used)",
                       "arithmetic": ("fp32 features and accumulation; matrix products on the 16-bit MFMA pipe: gathered 3x3x3 "
                                      "convolutions from an exact 3-way bf16 split of both operands (6 terms, 24 bits); dense and "
                                      "pair-list products from row/column-scaled fp16 pairs (3 terms, >= 22 bits relative to the "
                                      "row / column maximum; worst-case bound and adversarial cases: DESIGN.md section 4b, "
                                      "tests/test_gpu_fp16_pairs.py). `ms_per_step_strict` is the same step with every product "
                                      "on the six-term 24-bit form, `ms_per_step_fp32_mfma` on the fp32-input MFMA instructions")
                       if split_on else "fp32-input MFMA",
                       "ms_per_step_strict": strict_ms, "ms_per_step_fp32_mfma": fp32_ms,
                       "decode_ms_reusing_encoder_sets": cached_dec_ms,
                       "bpp_y_z_strings": bpp, "bpp_likelihood": rd["bpp_likelihood"], "d1_psnr": rd["d1_psnr_sym"],
                       "rate_distortion": rd, "ms_per_step_without_entropy_coder": hot_ms,
                       "frames_per_step": world, "encode_ms": recs[0][2] * 1e3, "decode_ms": recs[0][3] * 1e3,
                       "step_ms_rank0": [round((b - a) * 1e3, 2) for a, b in zip(step_marks[:-1], step_marks[1:])],
                       "per_rank": [{"rank": int(d[0]), "encode_ms": round(d[1], 3), "decode_ms": round(d[2], 3), "ms_per_step": round(d[3], 3),
                                     "step_ms_min": round(d[4], 3), "step_ms_max": round(d[5], 3), "host_cores": int(d[6]),
                                     "first_core": int(d[7])} for d in diag],
                       "device": arch, "cus": cu},
            "roofline": roof,
        }
        if rehearse:
            line["rehearsal"] = f"{world} ranks sharing ONE GPU, collectives over gloo: a control-flow check, not a measurement"
        if world == 1 and not args.no_aux:
            line["config"]["aux_true_geometry"] = true_geometry_ms(model, pc, q)
            line["config"]["aux_train_step"] = train_step_ms(device)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
