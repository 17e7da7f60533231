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


def step(model, pc, q):
    out = model.compress(pc, q, block_size=1024)
    rec = model.decompress(coordinates=out[3], strings=out[0], shape=out[1], k=out[2], q_vals=out[4])
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


def pmc_traffic():
    """HBM bytes per MFMA-conv launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.py; FETCH_SIZE and
    WRITE_SIZE in separate passes, KiB units, gfx950 FETCH x2 correction), with the hash of the kernel sources it was
    collected on.  (None, reason) when no measurement is committed or when it belongs to other kernel code."""
    try:
        rec = json.load(open(os.path.join(ROOT, PMC_FILE)))
    except Exception:
        return None, "no PMC measurement committed"
    now = csrc_sha()
    if rec.get("csrc_sha") != now:
        return None, f"stale: measured on csrc {rec.get('csrc_sha')}, running csrc {now} (re-run tools/pmc_traffic.py)"
    return float(rec["hbm_bytes_per_launch"]), f"rocprofv3 PMC passes of `{rec.get('command')}` on csrc {now}, {PMC_FILE}"


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

    def spy_r(feats, packed_w, bias, K, cin, cout, csr, n_out, act=0, slope=0.01):
        calls.append((int(csr[0][n_out].item()), float(K), cin, cout, n_out, feats.shape[0]))
        return orig_r(feats, packed_w, bias, K, cin, cout, csr, n_out, act, slope)

    def spy_h(feats, packed_w0, bias0, cmid, w2, bias2, cset, kmap):    # fused head: the cin -> cmid convolution is the MFMA launch
        calls.append((kmap, 27, feats.shape[1], cmid, feats.shape[0], feats.shape[0]))
        return orig_h(feats, packed_w0, bias0, cmid, w2, bias2, cset, kmap)

    orig_g = S.convt_forward_csr_grid

    def spy_g(feats, packed_w, bias, K, cin, cout, csr, out_set, act, ex_bias, slope=0.01):
        calls.append((int(csr[0][out_set.n].item()), -K, cin, cout, out_set.n, feats.shape[0]))
        return orig_g(feats, packed_w, bias, K, cin, cout, csr, out_set, act, ex_bias, slope)

    names = ("conv_forward", "convt_forward", "convt_forward_csr", "convt_forward_rows", "conv_head_forward",
             "convt_forward_csr_grid")
    S.COUNT_PAIRS = True
    for nme, f in zip(names, (spy, spy_t, spy_c, spy_r, spy_h, spy_g)):
        setattr(S, nme, f)
    try:
        step(model, pc, q)
        torch.cuda.synchronize()
    finally:
        S.COUNT_PAIRS = False
        for nme, f in zip(names, (orig, orig_t, orig_c, orig_r, orig_h, orig_g)):
            setattr(S, nme, f)
    flops, launches, pairs_total, alg_bytes, exec_flops = 0.0, 0, 0, 0.0, 0.0
    split_on = os.environ.get("PCC_MFMA_SPLIT", "1") != "0"
    h_on = os.environ.get("PCC_GEMM_H", "1") != "0"
    for kmap, K, cin, cout, n_out, n_in in calls:
        pair_rows = isinstance(K, float)                  # K given as a float: kept-row transposed convolution (pair GEMM)
        K = int(K)
        kmap_is_map = not isinstance(kmap, int) and kmap is not None
        dense, K = K < 0, abs(K)                          # K < 0: dense products of a generative transposed convolution
        if not mfma_shape(cin, cout):
            continue
        p = kmap if isinstance(kmap, int) else (kmap.pairs() if kmap is not None else n_out)
        fl = 2.0 * p * cin * cout
        flops += fl
        # 16-bit MFMA FLOPs executed per algorithmic FLOP: 3 (dense products, scaled fp16 pairs: k_gemm_h2), 6 (bf16 split:
        # k_conv_mfma_bf), or the fp32-input MFMA kernels priced at their own peak (2500 / 157.3)
        if cin % 32 != 0 or not split_on:
            terms = PEAK_BF16_MFMA_TFLOPS / PEAK_FP32_MFMA_TFLOPS
        elif (dense and h_on and cin <= 256 and cin // 32 in (1, 2, 4, 6, 8) and K * cout >= 128 and
              (-(-n_in // 128) + 7) // 8 * 8 * (-(-(K * cout) // 128)) >= 512):
            terms = 3.0
        elif (h_on and K >= max(64, S.PAIR_MIN_K) and cin >= S.PAIR_MIN_CIN and cin <= 256 and cin // 32 in (1, 2, 4, 6, 8)
              and cout >= 128 and cout % 4 == 0 and (pair_rows or kmap_is_map)):
            terms = 3.0                                   # gathered pair GEMM (k_pair_h2)
        else:
            terms = float(SPLIT_TERMS)
        exec_flops += fl * terms
        # compulsory traffic of the layer: every feature row, weight, map entry, coordinate touched once
        alg_bytes += 4.0 * (n_in * cin + n_out * cout + K * cin * cout) + 8.0 * p + 16.0 * (n_in + n_out)
        pairs_total += p
        launches += 1
    return flops, launches, pairs_total, alg_bytes, exec_flops


def cpu_baseline(threads=None):
    """The oracle (numpy restatement, 'port'; MinkowskiEngine cannot run here) timed on this host's cores on bounded
    samples of the same workload: the same synthetic surface at vox8 (1 warm-up, median of 3) and once at vox9, same
    R2 architecture and weights recipe, encode + decode, harness of BASELINE.md section 3.  `value` is the vox9 figure
    (closest to the benchmark's frame that fits the time bound); the full vox10 frame takes the oracle ~150 s on the
    build container's 8 cores (5.2 k points/s, tests/golden/make_fullsize.py), see DESIGN.md section 5."""
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

    pc8, pc9 = synth.surface_cloud(0, 8), synth.surface_cloud(0, 9)
    once(pc8)
    t8 = sorted(once(pc8) for _ in range(3))[1]
    t9 = once(pc9)
    if limit is not None:
        limit.restore_original_limits()
    return {"value": pc9.shape[0] / t9, "unit": "points/s", "cores": threads, "kind": "port",
            "points_per_s_vox8": pc8.shape[0] / t8,
            "sample": f"oracle (numpy restatement; MinkowskiEngine unavailable) encode+decode of the same synthetic surface, "
                      f"R2 architecture: vox9 ({pc9.shape[0]} points) one run {t9:.1f} s -> value; vox8 ({pc8.shape[0]} points) "
                      f"1 warm-up + median of 3 = {t8:.1f} s"}


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
        return model.decompress(coordinates=out[3], strings=out[0], shape=out[1], k=out[2], q_vals=out[4], probe=probe)

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


def train_step_ms(device, steps=5, warmup=2):
    """Auxiliary (BASELINE configs[3], never `value`): one training step -- forward, losses, backward, gradient clipping,
    Adam -- on 4 cubes of 128^3 cut from the benchmark frame, `configs/CVPR_inverse_scaling.yaml` (adaptive bottleneck,
    quantisation offsets, inverse rescaling, STE), as `train.py:178-240` runs it."""
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
    opt = torch.optim.Adam([p for nme, p in model.named_parameters() if not nme.endswith(".quantiles")], lr=1e-4)
    loss_fn = Loss(copy.deepcopy(loss_cfg))
    coords, feats = coords.to(device), feats.float().to(device)

    def one():
        x = ME.SparseTensor(coordinates=coords, features=feats)
        opt.zero_grad(set_to_none=True)
        out = model(x, q, Lam)
        total, _ = loss_fn(x, out)
        total.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        return float(total.detach())

    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        last = one()
    torch.cuda.synchronize()
    return {"train_step_ms": (time.time() - t0) / steps * 1e3, "cubes": nb, "points": int(coords.shape[0]),
            "loss": last, "config": "CVPR_inverse_scaling (adaptive_BN, offsets, inverse rescaling, STE), R2 width"}


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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

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
            dist.barrier(device_ids=[local_rank])

    for _ in range(args.warmup):
        step(model, pc, q)
    torch.cuda.synchronize()
    flops_step, launches_step, pairs_step, alg_bytes_step, exec_flops_step = (account_flops(model, pc, q) if rank == 0
                                                                              else (0.0, 0, 0, 0.0, 0.0))

    lib.call("pcc_prof_enable", 1 if rank == 0 else 0)
    barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    t_enc = 0.0
    step_marks = []
    for _ in range(args.steps):
        te = time.time()
        out = model.compress(pc, q, block_size=1024)
        torch.cuda.synchronize()
        t_enc += time.time() - te
        rec = model.decompress(coordinates=out[3], strings=out[0], shape=out[1], k=out[2], q_vals=out[4])
        step_marks.append(time.time())          # host time only (no sync): the next compress starts with a size read
    torch.cuda.synchronize()
    barrier()
    dt = time.time() - t0
    import ctypes as C
    conv_ms, conv_launches = C.c_double(0), C.c_int64(0)
    if rank == 0:
        lib.check(lib.load().pcc_prof_collect(C.byref(conv_ms), C.byref(conv_launches)), "pcc_prof_collect")
    lib.call("pcc_prof_enable", 0)

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

    tt = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt_max = float(tt.item())
    recs = frames.gather_records([(rank, n_points, t_enc / args.steps, (dt - t_enc) / args.steps, 0.0, rec.shape[0])],
                                 device)
    total_points = sum(r[1] for r in recs)

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
        traffic, traffic_note = pmc_traffic()
        split_on = os.environ.get("PCC_MFMA_SPLIT", "1") != "0"
        # roof of the ALGORITHMIC FLOPs of this mix of launches: the dense 16-bit MFMA peak divided by the 16-bit MFMA FLOPs
        # executed per algorithmic FLOP (3 for the dense products, 6 for the gathered convolutions, see account_flops)
        peak = PEAK_BF16_MFMA_TFLOPS * flops_step / exec_flops_step if exec_flops_step else PEAK_FP32_MFMA_TFLOPS
        ms_step = dt_max / args.steps * 1e3
        # conv launches: only MFMA-shaped ones are event-timed inside the library
        ach = (flops_step * args.steps / (conv_ms.value * 1e-3) / 1e12) if conv_ms.value > 0 else None
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
                                   f"1 block; entropy coder in the timed region: {args.coder}",
                       "arithmetic": ("fp32 features and accumulation; matrix products on the 16-bit MFMA pipe at fp32 accuracy: "
                                      "gathered convolutions from an exact 3-way bf16 split of both operands (6 terms, "
                                      "tests/test_gpu_map_conv.py::test_split_path_accuracy), dense products of the generative "
                                      "convolutions from row/column-scaled fp16 pairs (3 terms, ::test_dense_products_accuracy)")
                       if split_on else "fp32-input MFMA",
                       "bpp_y_z_strings": bpp, "bpp_likelihood": rd["bpp_likelihood"], "d1_psnr": rd["d1_psnr_sym"],
                       "rate_distortion": rd, "ms_per_step_without_entropy_coder": hot_ms,
                       "frames_per_step": world, "encode_ms": recs[0][2] * 1e3, "decode_ms": recs[0][3] * 1e3,
                       "step_ms_rank0": [round((b - a) * 1e3, 2) for a, b in zip([t0] + step_marks[:-1], step_marks)],
                       "device": arch, "cus": cu},
            # dominant kernel family: the MFMA convolutions (k_conv_mfma_bf = fp32 products as 6 exact bf16 MFMA terms).
            # `achieved` = ALGORITHMIC fp32 FLOPs (2*P*Cin*Cout) / event-timed launch durations; `peak` = the roof of that
            # arithmetic for algorithmic FLOPs, i.e. the dense bf16 MFMA peak / 6 terms.  (The fp32-input MFMA pipe used in
            # round 1 peaks at 157.3 TFLOP/s: `vs_fp32_mfma_peak`.)
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                         "frac": (ach / peak) if ach else None, "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_note,
                         "alg_bytes_per_launch": (alg_bytes_step / launches_step) if launches_step else None,
                         "kernel": "k_gemm_h2 (dense products of pcc_convt_fwd_csr*, 3 fp16 MFMA terms) + k_conv_mfma_bf (pcc_conv_fwd / "
                                   "pcc_conv_fwd_pairs, 6 bf16 MFMA terms)" if split_on else "k_conv_mfma (fp32-input MFMA)",
                         "peak_note": ("dense 16-bit MFMA peak 2500 TFLOP/s / executed 16-bit FLOPs per algorithmic FLOP "
                                       f"({exec_flops_step / flops_step:.2f} for this mix of launches)" if (split_on and flops_step)
                                       else "dense fp32-input MFMA peak"),
                         "executed_16bit_tflops": (ach * exec_flops_step / flops_step) if (ach and flops_step) else None,
                         "vs_fp32_mfma_peak": (ach / PEAK_FP32_MFMA_TFLOPS) if ach else None,
                         "flop_per_step": flops_step, "pairs_per_step": pairs_step, "launches_per_step": launches_step,
                         "avg_launch_ms": (conv_ms.value / conv_launches.value) if conv_launches.value else None,
                         "conv_ms_per_step": conv_ms.value / args.steps},
        }
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
