"""The three-term fp16 form of the dense / pair-list products (`k_gemm_h2`, `k_pair_h2`): worst-case bound, adversarial
operands, and the range guard.

The form: per feature row i and weight column j a power-of-two scale brings the largest magnitude to [2^14, 2^15); every
element is carried as h + l with h = fp16(s x), l = fp16(s x - h); a product is h h' + h l' + l h' on the fp16 MFMA (exact
products, fp32 accumulation), un-scaled on the way out.  An element is therefore exact to 2^-22 of ITS OWN magnitude while
its residual l is a normal fp16 number (|x| >= 2^-18 of the row maximum) and to an ABSOLUTE 2^-28 of the row maximum below
that (2^-39 if the matrix pipe keeps fp16 subnormals; the bound below assumes it flushes them).  Against exact arithmetic:

    |T^_ij - T_ij|  <=  (3 * 2^-22 + (cin + 2) * 2^-24) * S_ij  +  cin * 2^-27 * max_i * max_j,
    S_ij = sum_c |x_ic| |w_cj|,   max_i = max_c |x_ic|,   max_j = max_c |w_cj|

The first term is what fp32 arithmetic itself admits (plus the dropped l l' and the two representation errors); the second
is the price of the shared exponent.  The range guard watches the second term: a launch whose scales admit more than
`lib.H_GUARD_BUDGET` sets a device flag and the codec repeats the call in the six-term bf16 form (24 bits per element).
Every assertion here is ELEMENT-WISE, against float64 and against the fp32-input MFMA path.
"""
import numpy as np
import pytest
import torch

from tests.util import dev, t, n

pytestmark = pytest.mark.gpu

CIN = 128


def _dense(x, w, arithmetic):
    """x [rows, cin] @ w [cin, ncol] through pcc_convt_fwd_csr's dense product (one pair per output row: the per-pair buffer
    itself comes back).  arithmetic: "h" three-term fp16, "bf" six-term bf16, "f32" fp32-input MFMA."""
    from unified_point_cloud_compression_amd import sparse as S, lib as L
    rows, cin = x.shape
    ncol = w.shape[1]
    K = 8
    cout = ncol // K
    W = torch.nn.Parameter(t(np.ascontiguousarray(w.reshape(cin, K, cout).transpose(1, 0, 2))))
    first = torch.arange(0, rows * K + 1, dtype=torch.int32, device=dev())
    pair_ids = torch.arange(0, rows * K, dtype=torch.int32, device=dev())
    form = {"h": L.ARITH_H3, "bf": L.ARITH_BF6, "f32": L.ARITH_F32}[arithmetic]
    with L.arith_scope(form):
        pk = S.PackedConv(True).get(W)
        got = S.convt_forward_csr(t(x), pk, None, K, cin, cout, (first, pair_ids), rows * K)
    return n(got).reshape(rows, ncol).astype(np.float64)


def _bound(x, w):
    x64, w64 = np.abs(x.astype(np.float64)), np.abs(w.astype(np.float64))
    S = x64 @ w64
    cin = x.shape[1]
    return (3 * 2.0 ** -22 + (cin + 2) * 2.0 ** -24) * S + cin * 2.0 ** -27 * x64.max(1, keepdims=True) * w64.max(0, keepdims=True)


def _check(x, w, what):
    want = x.astype(np.float64) @ w.astype(np.float64)
    B = _bound(x, w)
    h, f32 = _dense(x, w, "h"), _dense(x, w, "f32")
    eh, ef = np.abs(h - want), np.abs(f32 - want)
    worst = (eh / np.maximum(B, 1e-300)).max()
    print(f"{what}: max |h - exact| / bound = {worst:.3f}; max |h - exact| = {eh.max():.3e}; max |fp32 path - exact| = {ef.max():.3e}; "
          f"max |h - fp32 path| = {np.abs(h - f32).max():.3e}")
    assert (eh <= B).all(), f"{what}: {int((eh > B).sum())} elements above the worst-case bound (worst ratio {worst:.2f})"
    # against the fp32-input MFMA path, element by element: both sit inside their own bounds around the exact product
    assert (np.abs(h - f32) <= B + (x.shape[1] + 2) * 2.0 ** -24 * (np.abs(x.astype(np.float64)) @ np.abs(w.astype(np.float64)))).all()
    return h, want, B


ROWS, NCOL = 4096 + 5, 2048


def test_plain_and_spread_operands_meet_the_bound_elementwise():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((ROWS, CIN)).astype(np.float32)
    w = (rng.standard_normal((CIN, NCOL)) / np.sqrt(CIN)).astype(np.float32)
    _check(x, w, "normal data")
    _check((x * np.exp(rng.standard_normal(x.shape) * 4)).astype(np.float32), (w * np.exp(rng.standard_normal(w.shape) * 3)).astype(np.float32),
           "magnitudes spread over e^+-12 element by element")


def test_one_element_2_pow_20_above_the_rest_of_its_row():
    """The adversarial row: one channel 2^20 times the others, and weight columns that IGNORE that channel, so the output is
    made of the small elements only -- they are carried with an absolute error of 2^-28 of the big one.  The bound's second
    term is exactly this case; the result must stay inside it (and inside 1e-4 for operands of this size)."""
    rng = np.random.default_rng(1)
    x = rng.standard_normal((ROWS, CIN)).astype(np.float32) * 2.0 ** -10
    big = rng.integers(0, CIN, ROWS)
    x[np.arange(ROWS), big] = 2.0 ** 10 * np.sign(rng.standard_normal(ROWS)).astype(np.float32)
    w = (rng.standard_normal((CIN, NCOL)) / np.sqrt(CIN)).astype(np.float32)
    w[:, ::2] *= (rng.random((CIN, NCOL // 2)) < 0.5)                      # half the columns miss half the channels
    for c in range(0, NCOL, 4):
        w[:, c] = np.where(np.arange(CIN) % 7 == 0, 0.0, w[:, c])
    h, want, B = _check(x, w, "one element 2^20 above its row")
    # what the shared exponent costs here, in absolute terms: max_i = 2^10, max_j ~ 0.3 -> cin * 2^-27 * 2^10 * 0.3 = 3e-4 admitted,
    # measured far lower (the bound adds |errors|; they are signed and the residual of a small element is rarely at its worst)
    small_cols = np.abs(want) < 1.0
    print(f"   outputs made of small elements: {int(small_cols.sum())}; their max abs error {np.abs(h - want)[small_cols].max():.3e}")


def test_cancelling_pairs():
    """(x, -x + eps) pairs against equal weights: T is ~2^-12 of S.  The error stays bounded relative to S (as fp32's does),
    not relative to T."""
    rng = np.random.default_rng(2)
    v = rng.standard_normal((ROWS, CIN // 2)).astype(np.float32)
    x = np.empty((ROWS, CIN), np.float32)
    x[:, 0::2] = v
    x[:, 1::2] = -v * np.float32(1 - 2.0 ** -12)
    wv = (rng.standard_normal((CIN // 2, NCOL)) / np.sqrt(CIN)).astype(np.float32)
    w = np.repeat(wv, 2, axis=0)
    _check(x, w, "cancelling pairs")


def test_columns_whose_small_weights_fall_below_fp16_normal_range_after_scaling():
    """A weight column with one entry 2^31 times the others: scaled, the small ones land below 2^-14, the smallest normal fp16
    number (sub-normal h, no l at all).  Rows that are zero in the big entry's channel see only those.  Also reports whether
    the matrix pipe kept the sub-normal inputs (error at the 2^-39 level) or flushed them (2^-28 level): the bound holds either way."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((ROWS, CIN)).astype(np.float32)
    w = (rng.standard_normal((CIN, NCOL)) * 2.0 ** -24).astype(np.float32)
    bigc = rng.integers(0, CIN, NCOL)
    w[bigc, np.arange(NCOL)] = 2.0 ** 7
    x[::2, :] = np.where(np.isin(np.arange(CIN), np.unique(bigc[:64])), 0.0, x[::2, :])     # half the rows miss the big channels of the first columns
    h, want, B = _check(x, w, "columns with sub-normal-after-scaling entries")
    sel = (np.abs(want) < 1e-3)
    if sel.any():
        rel_floor = np.abs(h - want)[sel].max() / (2.0 ** 7 * np.abs(x).max())
        print(f"   outputs made of sub-normal-after-scaling weights: {int(sel.sum())}; max abs error / (max_i * max_j) = {rel_floor:.3e} "
              f"({'sub-normal fp16 inputs kept by the matrix pipe' if rel_floor < 2.0 ** -32 else 'consistent with flushed sub-normal inputs'})")


def test_range_guard_trips_exactly_when_the_scales_admit_more_than_the_budget():
    from unified_point_cloud_compression_amd import lib as L
    rng = np.random.default_rng(4)
    guard = L.h_guard(dev())
    guard.zero_()
    x = rng.standard_normal((ROWS, CIN)).astype(np.float32)            # (enough tiles for the dense products to take the three-term kernel)
    w = (rng.standard_normal((CIN, NCOL)) / np.sqrt(CIN)).astype(np.float32)
    _dense(x, w, "h")
    assert int(guard.item()) == 0                           # max_i ~ 4, max_j ~ 0.4: cin * 2^-27 * 1.6 = 1.5e-6 << budget
    x[17, 5] = 2.0 ** 12                                    # one row whose maximum times the column maxima admits 128 * 2^-27 * 4096 * 0.4 = 1.6e-3
    _dense(x, w, "h")
    assert int(guard.item()) == 1
    guard.zero_()
    _dense(x, w, "bf")                                      # the six-term form has no range condition and never touches the guard
    assert int(guard.item()) == 0


def _spread_model(coder="pcc_streams"):
    """R2 architecture, random weights, with the last hyper-synthesis layer rescaled so that the predicted scales spread over
    the whole scale table (random weights alone leave every sigma below the 0.11 bound: all indexes 0, nothing could flip)."""
    import copy
    from oracle import codec
    from tests.util import load_params
    from unified_point_cloud_compression_amd.model import UnifiedModel
    cfg = copy.deepcopy(codec.R2_CONFIG)
    P = codec.random_params(cfg, 0, gain=3.0)
    mcfg = copy.deepcopy(cfg)
    mcfg["entropy_model"]["entropy_coder"] = coder
    model = load_params(UnifiedModel(mcfg), P).to(dev()).eval()
    with torch.no_grad():
        last = model.entropy_model.h_s[4]
        c = last.out_channels // 2
        last.kernel[..., :c].mul_(3.0e2)                     # scales half: |sigma| from ~0 to a few tens
        last.bias[..., :c].add_(0.5)
    return model


class _Tap:
    """Records what the conditional model saw on each side: SHA of scales|means, the table indexes, the symbols."""

    def __init__(self, model):
        import hashlib
        self.gc = model.entropy_model.gaussian_conditional
        self.enc, self.dec = {}, {}
        gc, sha = self.gc, lambda x: hashlib.sha256(x.detach().cpu().numpy().tobytes()).hexdigest()
        self._orig = (gc.encode_rows, gc.index_rows, gc.decompress_rows)

        def encode_rows(y, params, keys=None, gain=None, **kw):
            out = self._orig[0](y, params, keys, gain, **kw)
            if y is not None:                                # (index_rows goes through encode_rows with y = None)
                self.enc = {"params": sha(params), "idx": out[1].clone(), "sym": out[0].clone()}
            return out

        def index_rows(params, keys=None, gain=None):
            idx = self._orig[1](params, keys, gain)
            self.dec["params"], self.dec["idx"] = sha(params), idx.clone()
            return idx

        def decompress_rows(*a, **kw):
            sym = self._orig[2](*a, **kw)
            self.dec["sym"] = sym.clone()
            return sym
        gc.encode_rows, gc.index_rows, gc.decompress_rows = encode_rows, index_rows, decompress_rows

    def close(self):
        self.gc.encode_rows, self.gc.index_rows, self.gc.decompress_rows = self._orig


def _count_fallbacks(L):
    """Spy on `arith_scope`: how many un-pinned six-term scopes (= range-guard fallbacks) were entered."""
    seen = []
    orig = L.arith_scope.__enter__

    def enter(self):
        if self.form == L.ARITH_BF6 and not self.pinned:
            seen.append(1)
        return orig(self)
    L.arith_scope.__enter__ = enter
    return seen, lambda: setattr(L.arith_scope, "__enter__", orig)


@pytest.mark.parametrize("side", ["encoder", "decoder", "neither"])
def test_guard_trip_on_one_side_only_keeps_encoder_and_decoder_in_step(side):
    """VERDICT r3 item 1.  The range-guard fallback re-runs g_a (encoder) or g_s (decoder) in the six-term form; the
    hyper-synthesis, whose scales / means select the rANS table rows on BOTH sides, is pinned to one form, so a trip on one
    side only cannot desynchronise the streams.  Weights chosen so that the indexes span >= 30 table rows; byte strings go
    through the GPU stream coder; the decoder must see the encoder's scales|means bit for bit (SHA), the same indexes and
    decode the encoder's symbols.  Reference contract: `model/entropy_models.py:371-400,438-484`."""
    from unified_point_cloud_compression_amd import lib as L, synth
    model = _spread_model()
    with torch.no_grad():
        if side == "encoder":
            model.g_a.down_conv_2[0].kernel.mul_(4096.0)         # pair-list 5x5x5 products: max|row| * max|column| far above 26
            model.g_a.down_conv_2[0].bias.mul_(4096.0)
            model.g_a.down_conv_2[1].beta.mul_(64.0)             # GDN: beta' = 4096 beta (beta = param^2 - 2^-36), so that
            #                                                      x' / (beta' + gamma |x'|) = x / (beta + gamma |x|): the
            #                                                      latents, hence the decoder's operands, keep their range
        elif side == "decoder":
            model.g_s.up_2[1].kernel.mul_(1024.0)                # level-2 up-sampling (today's case): ~100 > 26
            model.g_s.predict_2[0].kernel.mul_(1.0 / 1024.0)     # (keeps the level's logits in range)
    model.update()
    pc = torch.from_numpy(synth.surface_cloud(0, 8)).to(dev())
    q = torch.tensor([[0.5, 0.5]], device=dev())
    tap = _Tap(model)
    seen, restore = _count_fallbacks(L)
    try:
        out = model.compress(pc, q)
        enc_fallbacks = len(seen)
        enc = dict(tap.enc)
        rec = model.decompress(coordinates=[c.clone() for c in out[3]], strings=out[0], shape=out[1], k=out[2], q_vals=out[4])
        dec_fallbacks = len(seen) - enc_fallbacks
        # the same decode with EVERY product of g_s forced to the six-term form: what the fallback must reproduce bit for bit
        with L.arith_scope(L.ARITH_BF6):
            ref = model.decompress(coordinates=[c.clone() for c in out[3]], strings=out[0], shape=out[1], k=out[2], q_vals=out[4])
    finally:
        restore()
        tap.close()
    assert (enc_fallbacks, dec_fallbacks) == {"encoder": (1, 0), "decoder": (0, 1), "neither": (0, 0)}[side], (enc_fallbacks, dec_fallbacks)
    rows = torch.unique(enc["idx"]).numel()
    assert rows >= 30, f"indexes span only {rows} table rows: the test cannot see a flipped index"
    assert enc["params"] == tap.dec["params"], "scales|means differ between encoder and decoder"
    assert torch.equal(enc["idx"], tap.dec["idx"])
    assert torch.equal(enc["sym"], tap.dec["sym"]), "the decoder did not reproduce the encoder's y symbols"
    if side == "decoder":
        assert torch.equal(rec, ref)                             # fallback == a six-term run of g_s, bit for bit
    assert int(L.h_guard(dev()).item()) == 0                     # the guard word is left clean for the next call


def test_hyper_synthesis_bits_do_not_depend_on_the_ambient_form():
    """The pinned scope: scales|means from `_gaussian_params` are the same bits under every ambient form a caller may be in
    (three-term default, six-term fallback) -- only the diagnostic process-wide override ARITH_FORCE changes them, on both
    sides at once."""
    import hashlib
    from unified_point_cloud_compression_amd import lib as L, synth
    model = _spread_model("symbols")
    model.update()
    pc = torch.from_numpy(synth.surface_cloud(1, 8)).to(dev())
    q = torch.tensor([[0.5, 0.5]], device=dev())
    tap = _Tap(model)
    try:
        shas = []
        for form in (L.ARITH_H3, L.ARITH_BF6, L.ARITH_F32):
            with L.arith_scope(form):
                model.compress(pc, q)
            shas.append(tap.enc["params"])
    finally:
        tap.close()
    assert shas[0] == shas[1] == shas[2], shas


def test_real_activations_level_by_level_against_the_fp32_input_path():
    """The decoder of the R2 architecture on a real (synthetic-surface) frame, every composite level, three-term fp16 form
    against the fp32-input MFMA path: logits of every candidate element by element, on the activations the codec really
    produces (post-IGDN features, composite 7x7x7 weights)."""
    import copy
    from oracle import codec
    from tests.util import load_params
    from unified_point_cloud_compression_amd import lib as L, synth
    from unified_point_cloud_compression_amd.model import UnifiedModel
    cfg = copy.deepcopy(codec.R2_CONFIG)
    P = codec.random_params(cfg, 0, gain=3.0)
    mcfg = copy.deepcopy(cfg)
    mcfg["entropy_model"]["entropy_coder"] = "symbols"
    model = load_params(UnifiedModel(mcfg), P).to(dev()).eval()
    model.update()
    pc = torch.from_numpy(synth.surface_cloud(0, 9)).to(dev())
    q = torch.tensor([[0.5, 0.5]], device=dev())
    out = model.compress(pc, q)
    logits = {}

    def run(tag):
        seen = logits.setdefault(tag, {})
        forced = logits.get("h")

        def probe(stage, lvl, cset, logit, mask, feats):
            if stage != "select":
                return None
            seen[lvl] = (logit.detach().clone(), mask.clone())
            return forced[lvl][1] if (forced is not None and tag != "h") else None      # same kept set on both runs: same inputs downstream
        model.decompress(coordinates=[c.clone() for c in out[3]], strings=out[0], shape=out[1], k=out[2], q_vals=out[4], probe=probe)

    run("h")
    L.ARITH_FORCE = L.ARITH_F32                              # every product of the process, the pinned hyper-synthesis included
    try:
        run("f32")
    finally:
        L.ARITH_FORCE = None
    for lvl in range(3):
        a, b = logits["h"][lvl][0].double(), logits["f32"][lvl][0].double()
        assert a.shape == b.shape
        d = (a - b).abs()
        print(f"level {lvl}: {a.shape[0]} candidates, |logit| up to {b.abs().max().item():.2f}, max |three-term - fp32 path| = {d.max().item():.3e}")
        assert (d <= 2e-5 + 2e-5 * b.abs()).all()             # a fifth of the 1e-4 bar, element-wise
