"""GPU suite, module level: the drop-in nn.Modules (hri_emo_amd.models.*) against
  (1) the golden vectors generated from the reference import (tests/golden/*.npz), and
  (2) the CPU oracle on seeded inputs,
plus size-independent properties at the full BASELINE cfg-2 size.

Tolerance: north_star asks 1e-2 for the bf16 path; with the fp32 twin of the residual stream the measured
error is ~2e-3, so the tests hold outputs to |got - ref| <= 5e-3 * max(1, max|ref|);
gradients: per-parameter relative L2 error bounded by the reference path's own bf16 (autocast) error, see
test_fusion_train_step_grads.
"""
import os

import pytest
import torch

from conftest import load_golden
from oracle import hri_emo_oracle as O          # the checker (tests only)

pytestmark = pytest.mark.gpu
TOL = 5e-3


@pytest.fixture(scope="module")
def H():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import hri_emo_amd
    return hri_emo_amd


def close(got, ref, tol=TOL, what=""):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), (what, err, ref.abs().max().item())


def cu(t):
    return None if t is None else t.cuda()


def fusion(H, d, ne, p=0.1):
    return O.closed_form_init_(H.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=p)).cuda()


def test_state_dict_keys_and_strict_load(H):
    ref = O.closed_form_init_(O.FusionWithEmotionDecoder(d_model=128, num_emotions=4))
    m = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4)
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
    m.load_state_dict(ref.state_dict(), strict=True)


@pytest.mark.parametrize("name", ["cfg1_eval_nomask", "cfg1_eval_ragged", "cfg1_eval_2d_inputs"])
def test_fusion_eval_vs_golden(H, name):
    g = load_golden(name)
    m = fusion(H, 128, 4).eval()
    with torch.no_grad():
        logits, beta, z = m(cu(g["h_a"]), cu(g["h_t"]), cu(g.get("mask_a")), cu(g.get("mask_t")))
    assert logits.dtype == torch.float32 and z.dtype == torch.float32
    close(logits, g["logits"], what="logits"); close(beta, g["beta"], what="beta"); close(z, g["z"], what="z")


@pytest.mark.parametrize("name,d,ne", [("cfg1_eval_ragged", 128, 4), ("hd96_eval_ragged", 768, 6)])
def test_fusion_attention_maps_vs_golden(H, name, d, ne):
    g = load_golden(name)
    m = fusion(H, d, ne).eval()
    with torch.no_grad():
        logits, beta, z, pack = m(cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]), return_attention=True)
    close(logits, g["logits"], what="logits"); close(z, g["z"], what="z"); close(beta, g["beta"], what="beta")
    assert len(pack["encoder"]) == 2 and len(pack["decoder"]) == 2
    # maps are probabilities; with the fixture's peaked softmaxes (max p ~ 0.9999, |scores| ~ 20) the bf16
    # rounding of q/k moves a probability by up to p(1-p)*|ds| ~ 1.1e-2 -> 2e-2 absolute here
    for li, maps in enumerate(pack["encoder"]):
        for k, v in maps.items():
            close(v, g[f"enc.{li}.{k}"], 2e-2, what=f"enc.{li}.{k}")
    for li, v in enumerate(pack["decoder"]):
        close(v, g[f"dec.{li}"], 2e-2, what=f"dec.{li}")
    w = pack["encoder"][-1]["audio_queries_text"].cpu()
    assert (w[g["mask_t"][:, None, :].expand_as(w)] == 0).all()          # PAD key columns exactly 0
    close(w.sum(-1), torch.ones(w.shape[:-1]), 5e-3, "rows sum to one")


def test_attention_maps_default_init_within_1e_2(H):
    """north_star's bf16 tolerance (1e-2) on the exported attention maps, shown on a benign case: default torch init (softmax
    not saturated), d=768 / head_dim 96, ragged masks; HIP maps vs the fp32 oracle's, every layer, every map."""
    torch.manual_seed(1234)
    kw = dict(d_model=768, num_emotions=6, n_heads=8, dropout=0.1)
    ref = O.FusionWithEmotionDecoder(**kw).eval()
    m = H.FusionWithEmotionDecoder(**kw)
    m.load_state_dict(ref.state_dict())
    m.cuda().eval()
    h_a, h_t, m_a, m_t = _rand_batch(2, 100, 40, 768, 17)
    with torch.no_grad():
        lr, br, zr, pr = ref(h_a, h_t, m_a, m_t, return_attention=True)
        lg, bg, zg, pg = m(cu(h_a), cu(h_t), cu(m_a), cu(m_t), return_attention=True)
    close(lg, lr, what="logits"); close(zg, zr, what="z")
    worst = 0.0
    for li, (mg, mr) in enumerate(zip(pg["encoder"], pr["encoder"])):
        for k in mr:
            worst = max(worst, (mg[k].float().cpu() - mr[k]).abs().max().item())
            close(mg[k], mr[k], 1e-2, what=f"enc.{li}.{k}")
    for li, (vg, vr) in enumerate(zip(pg["decoder"], pr["decoder"])):
        worst = max(worst, (vg.float().cpu() - vr).abs().max().item())
        close(vg, vr, 1e-2, what=f"dec.{li}")
    assert worst <= 1e-2, worst


def test_fusion_allpad_row_nan_only_for_that_sample(H):
    g = load_golden("cfg1_eval_allpad_row")
    m = fusion(H, 128, 4).eval()
    with torch.no_grad():
        logits, beta, z = m(cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]))
    logits = logits.cpu()
    assert torch.equal(torch.isnan(logits), torch.isnan(g["logits"]))
    ok = ~torch.isnan(g["logits"])
    close(logits[ok], g["logits"][ok])


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _train_step(model, h_a, h_t, m_a, m_t, y, autocast_cpu=False):
    h_a = h_a.clone().requires_grad_(True)
    h_t = h_t.clone().requires_grad_(True)
    if autocast_cpu:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            logits, beta, z = model(h_a, h_t, m_a, m_t)
        logits, beta = logits.float(), beta.float()
    else:
        logits, beta, z = model(h_a, h_t, m_a, m_t)
    loss = O.train_step_loss(logits, beta, y)
    model.zero_grad()
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    return loss.detach(), logits.detach(), h_a.grad, h_t.grad, grads


GRAD_FLOOR = 4e-2          # relative L2 error any bf16 backward may show on a well-conditioned parameter
GRAD_FACTOR = 3.0          # ... or this many times the reference path's own bf16 (CPU autocast) error on THAT parameter


def assert_per_parameter_grads(gm, gy, gr, exceptions=(), what=""):
    """EVERY parameter n: rel-L2(mine[n], fp32 oracle[n]) <= max(floor, GRAD_FACTOR * rel-L2(yardstick[n], fp32 oracle[n])),
    the yardstick being the reference path itself in bf16 (torch CPU autocast of the oracle on the same weights).
    floor = max(GRAD_FLOOR, 1.25 x the 90th percentile of the yardstick's own per-parameter errors): with well-conditioned (default
    init) weights that is 4e-2; with the ill-conditioned closed-form fixture weights the reference-in-bf16 itself is 5-9 %
    off on most parameters and the floor follows it.  No parameter is exempt unless named in `exceptions` (name -> bound)."""
    exceptions = dict(exceptions)

    def rel(a, b, n):
        # Packed in-projection bias [q | k | v]: the K third has an EXACT gradient of zero (softmax does not change when a
        # constant is added to every key's score, so sum_k dS[q,k] = 0 and sum_q dK = 0): every finite-precision path, the
        # reference-in-bf16 included, returns rounding noise there.  The relative error is therefore taken over the q and v
        # thirds, and the noise third is bounded in absolute terms below.
        if n.endswith("in_proj_bias"):
            d3 = a.numel() // 3
            return _rel(torch.cat([a[:d3], a[2 * d3:]]), torch.cat([b[:d3], b[2 * d3:]]))
        return _rel(a, b)

    for n in gr:
        if n.endswith("in_proj_bias"):
            d3 = gr[n].numel() // 3
            qv = torch.cat([gr[n][:d3], gr[n][2 * d3:]]).float().norm().item()
            km, ky = gm[n][d3:2 * d3].float().norm().item(), gy[n][d3:2 * d3].float().norm().item()
            assert km <= max(3.0 * ky, 0.1 * qv), (what, n, "noise in the zero-gradient K third", km, ky, qv)
    ys = sorted(rel(gy[n], gr[n], n) for n in gr)
    floor = max(GRAD_FLOOR, 1.25 * ys[len(ys) * 9 // 10])
    rows, bad = [], []
    for n in gr:
        em, ey = rel(gm[n], gr[n], n), rel(gy[n], gr[n], n)
        bound = exceptions.get(n, max(floor, GRAD_FACTOR * ey))
        rows.append((em, ey, n))
        if not em <= bound:
            bad.append((n, round(em, 4), round(ey, 4), round(bound, 4)))
    rows.sort(reverse=True)
    assert not bad, (what, "floor", floor, "parameters over their bound (name, mine, yardstick, bound):", bad, "worst five:", rows[:5])
    return rows


@pytest.mark.parametrize("name,d,ne,init", [("cfg1_train_p0", 128, 4, "closed"), ("hd96_train_p0", 768, 6, "closed"),
                                            ("cfg1_train_p0", 128, 4, "random"), ("hd96_train_p0", 768, 6, "random")])
def test_fusion_train_step_grads(H, name, d, ne, init):
    """fwd+bwd of the trainer's step (train_fusion_seq_level_decoder.py:310-331, dropout=0) on the golden
    inputs.  Gradient accuracy of a bf16 path depends on the conditioning of the weights, so the bound is
    the reference path itself evaluated in bf16: torch's CPU autocast(bfloat16) of the oracle on the same
    weights is the yardstick, and EVERY parameter's relative L2 error (vs the fp32 oracle) must stay within
    max(4e-2, 3x the yardstick's error on that same parameter) -- see assert_per_parameter_grads.  With
    the closed-form fixture weights the fp32 oracle's loss/logits/grad norms are additionally the
    committed golden values."""
    g = load_golden(name)
    torch.manual_seed(1234)
    ref = O.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=0.0).train()
    if init == "closed":
        O.closed_form_init_(ref)
    m = H.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=0.0)
    m.load_state_dict(ref.state_dict())
    m.cuda().train()
    loss_r, logits_r, ga_r, gt_r, gr = _train_step(ref, g["h_a"], g["h_t"], g["mask_a"], g["mask_t"], g["y"])
    _, _, ga_y, gt_y, gy = _train_step(ref, g["h_a"], g["h_t"], g["mask_a"], g["mask_t"], g["y"], autocast_cpu=True)
    loss_m, logits_m, ga_m, gt_m, gm = _train_step(m, cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]),
                                                   cu(g["y"]))
    if init == "closed":           # the live fp32 oracle IS the golden (pins what we compare against)
        close(loss_r.reshape(1), g["loss"], 1e-5, "oracle loss"); close(logits_r, g["logits"], 1e-5, "oracle logits")
        for n in gr:
            close(gr[n].norm().reshape(1), g["g.norm." + n], 1e-4, n)
    close(loss_m.reshape(1), loss_r.reshape(1), what="loss"); close(logits_m, logits_r, what="logits")
    for n, p in m.named_parameters():
        assert p.grad.dtype == torch.float32 and p.grad.shape == p.shape, n
    assert_per_parameter_grads(gm, gy, gr, what=f"{name}/{init}")
    assert _rel(ga_m, ga_r) <= max(3e-2, 1.5 * _rel(ga_y, ga_r)), "d loss / d h_a"
    assert _rel(gt_m, gt_r) <= max(3e-2, 1.5 * _rel(gt_y, gt_r)), "d loss / d h_t"


def test_fusion_train_step_grads_cfg2_shape(H):
    """The same per-parameter gradient bound at the HEADLINE shape (BASELINE configs[1]: d=768, T_a=400, T_t=128, N_e=6,
    H=8, 2+2 layers; B=2 keeps the CPU oracle and its autocast yardstick within seconds), default torch init, ragged masks."""
    torch.manual_seed(1234)
    kw = dict(d_model=768, num_emotions=6, n_heads=8, dropout=0.0)
    ref = O.FusionWithEmotionDecoder(**kw).train()
    m = H.FusionWithEmotionDecoder(**kw)
    m.load_state_dict(ref.state_dict())
    m.cuda().train()
    h_a, h_t, m_a, m_t = _rand_batch(2, 400, 128, 768, 41)
    y = (torch.rand(2, 6, generator=torch.Generator().manual_seed(42)) < 0.3).float()
    loss_r, logits_r, ga_r, gt_r, gr = _train_step(ref, h_a, h_t, m_a, m_t, y)
    _, _, ga_y, gt_y, gy = _train_step(ref, h_a, h_t, m_a, m_t, y, autocast_cpu=True)
    loss_m, logits_m, ga_m, gt_m, gm = _train_step(m, cu(h_a), cu(h_t), cu(m_a), cu(m_t), cu(y))
    close(loss_m.reshape(1), loss_r.reshape(1), what="loss"); close(logits_m, logits_r, what="logits")
    assert_per_parameter_grads(gm, gy, gr, what="cfg2 shape")
    assert _rel(ga_m, ga_r) <= max(3e-2, 1.5 * _rel(ga_y, ga_r)), "d loss / d h_a"
    assert _rel(gt_m, gt_r) <= max(3e-2, 1.5 * _rel(gt_y, gt_r)), "d loss / d h_t"


def test_cfg2_shape_vs_reference_golden(H):
    """the HIP path at the headline shape (d=768, T_a=400, T_t=128, N_e=6; B=2, ragged masks) against outputs, loss and gradient
    record generated from the reference itself (tests/golden/cfg2_seeded.npz; inputs regenerated from the seed)"""
    from conftest import cfg2_seeded_inputs
    g = load_golden("cfg2_seeded")
    h_a, h_t, m_a, m_t = cfg2_seeded_inputs(g)
    m = fusion(H, 768, 6, p=0.0).eval()
    with torch.no_grad():
        logits, beta, z = m(cu(h_a), cu(h_t), cu(m_a), cu(m_t))
    close(logits, g["logits"], what="logits"); close(beta, g["beta"], what="beta"); close(z, g["z"], what="z")
    # gradients: the fp32 oracle on the same (closed-form) weights is first tied to the reference's gradient record of the
    # fixture, then every parameter of the HIP step is held to the per-parameter bound against it (assert_per_parameter_grads:
    # floor / 3x the reference path's own bf16 error on that parameter)
    ref = O.closed_form_init_(O.FusionWithEmotionDecoder(d_model=768, num_emotions=6, n_heads=8, dropout=0.0)).train()
    loss_r, logits_r, ga_r, gt_r, gr = _train_step(ref, h_a, h_t, m_a, m_t, g["y"])
    close(loss_r.reshape(1), g["loss"], 1e-5, "oracle loss vs golden")
    for n in gr:
        close(gr[n].norm().reshape(1), g["g.norm." + n], 1e-4, "oracle grad norm vs golden " + n)
    _, _, ga_y, gt_y, gy = _train_step(ref, h_a, h_t, m_a, m_t, g["y"], autocast_cpu=True)
    m.train()
    loss_m, logits_m, ga_m, gt_m, gm = _train_step(m, cu(h_a), cu(h_t), cu(m_a), cu(m_t), cu(g["y"]))
    assert abs(float(loss_m) - float(g["loss"])) <= 2e-3 * max(1.0, abs(float(g["loss"])))
    gm = {n: v.float().cpu() for n, v in gm.items()}
    rows = assert_per_parameter_grads(gm, gy, gr, what="cfg2 shape vs golden-tied oracle")
    print("cfg-2 shape: worst five (mine, yardstick, parameter):", [(f"{a:.3f}", f"{b:.3f}", n) for a, b, n in rows[:5]])
    ea, ya = _rel(ga_m, ga_r), _rel(ga_y, ga_r)
    et, yt = _rel(gt_m, gt_r), _rel(gt_y, gt_r)
    assert ea <= max(GRAD_FLOOR, GRAD_FACTOR * ya) and et <= max(GRAD_FLOOR, GRAD_FACTOR * yt), (ea, ya, et, yt)


@pytest.mark.parametrize("name,seed,Ta,Tt,kw", [("cfg4_seeded", 31, 1000, 50, dict(d_model=768, num_emotions=6, n_heads=8)),
                                                ("cfg5_seeded", 41, 400, 128, dict(d_model=1024, num_emotions=7, n_heads=8,
                                                                                   num_layers_fusion=4, num_layers_decoder=2))])
def test_other_baseline_configs_vs_reference_golden(H, name, seed, Ta, Tt, kw):
    """BASELINE configs[3] (MOSEI shape) and configs[4]'s dimensions (bf16 GEMMs) against outputs generated from the reference"""
    from conftest import cfg2_seeded_inputs
    g = load_golden(name)
    h_a, h_t, m_a, m_t = cfg2_seeded_inputs(g, seed, Ta, Tt, kw["d_model"])
    m = O.closed_form_init_(H.FusionWithEmotionDecoder(dropout=0.0, **kw)).cuda().eval()
    with torch.no_grad():
        logits, beta, z = m(cu(h_a), cu(h_t), cu(m_a), cu(m_t))
    close(logits, g["logits"], what="logits"); close(beta, g["beta"], what="beta"); close(z, g["z"], what="z")


def test_components_vs_golden(H):
    g = load_golden("block_eval_ragged")
    blk = O.closed_form_init_(H.CrossModalBlock(128, 8, 0.1)).cuda().eval()
    with torch.no_grad():
        oa, ot, maps = blk(cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]), return_attention=True)
    assert oa.dtype == torch.float32
    close(oa, g["out_a"], what="out_a"); close(ot, g["out_t"], what="out_t")
    for k, v in maps.items():
        close(v, g["map." + k], 2e-2, what=k)            # probabilities: bf16 q/k rounding, see the maps test
    gg = load_golden("gate_eval_ragged")
    gate = O.closed_form_init_(H.BetaGate(128, 32)).cuda().eval()
    with torch.no_grad():
        hf, beta = gate(cu(gg["h_a"]), cu(gg["h_t"]), cu(gg["mask_a"]), cu(gg["mask_t"]))
    # h_fusion is produced in bf16 (it is the decoder's GEMM operand): one bf16 ulp at |x| in [2,4) is 1.6e-2,
    # so the stand-alone gate output is held to the north-star 1e-2 rather than the 5e-3 of the fp32-twin outputs
    close(hf, gg["h_fusion"], 1e-2, what="h_fusion"); close(beta, gg["beta"], what="beta")
    gg = load_golden("gate_eval_equal_len_nomask")
    with torch.no_grad():
        hf, beta = gate(cu(gg["h_a"]), cu(gg["h_t"]))
    close(hf, gg["h_fusion"], 1e-2, what="h_fusion eq"); close(beta, gg["beta"], what="beta eq")
    gd = load_golden("decoder_eval_ragged")
    dec = O.closed_form_init_(H.EmotionDecoder(128, 5, 8, 2, 64, 0.1)).cuda().eval()
    with torch.no_grad():
        z, logits, maps = dec(cu(gd["memory"]), cu(gd["mask"]), return_attention=True)
    # stand-alone decoder on the structured fixture weights (dim_feedforward=64): 6e-3 -> north-star 1e-2
    close(z, gd["z"], 1e-2, what="z"); close(logits, gd["logits"], 1e-2, what="logits")
    for i, v in enumerate(maps):
        close(v, gd[f"map.{i}"], 2e-2, what=f"dec map {i}")


def _rand_batch(B, Ta, Tt, d, seed, ragged=True):
    g = torch.Generator().manual_seed(seed)
    h_a, h_t = torch.randn(B, Ta, d, generator=g), torch.randn(B, Tt, d, generator=g)
    if not ragged:
        return h_a, h_t, None, None
    la = torch.randint(max(1, Ta // 2), Ta + 1, (B,), generator=g)
    lt = torch.randint(max(1, Tt // 2), Tt + 1, (B,), generator=g)
    return h_a, h_t, torch.arange(Ta)[None] >= la[:, None], torch.arange(Tt)[None] >= lt[:, None]


@pytest.mark.parametrize("B,Ta,Tt,d,ne,lf,ld", [
    (3, 100, 40, 768, 6, 2, 2), (2, 130, 50, 256, 7, 2, 2), (4, 32, 16, 128, 4, 2, 2),
    (2, 400, 128, 768, 6, 2, 2),          # BASELINE configs[1]/[2]: THE headline shape (IEMOCAP seq-level), per-GPU shard of cfg 3
    (2, 1000, 50, 768, 6, 2, 2),          # BASELINE configs[3]: MOSEI shape, long asymmetric cross-attention
    (2, 400, 128, 1024, 7, 4, 2),         # BASELINE configs[4] dimensions (d=1024 -> head_dim 128, 4+2 layers), bf16 path
    (2, 64, 64, 512, 5, 1, 1),            # L_a == L_t (no truncation in the gate), head_dim 64, 1+1 layers
    (1, 1, 1, 128, 1, 1, 1),              # one utterance, one frame, one token, one emotion query
    (3, 17, 5, 128, 3, 2, 2), (5, 33, 7, 256, 2, 1, 2),     # odd lengths
])
def test_fusion_vs_oracle_seeded(H, B, Ta, Tt, d, ne, lf, ld):
    torch.manual_seed(1234)
    kw = dict(d_model=d, num_emotions=ne, n_heads=8, dropout=0.1, num_layers_fusion=lf, num_layers_decoder=ld)
    ref = O.FusionWithEmotionDecoder(**kw)   # default torch init
    m = H.FusionWithEmotionDecoder(**kw)
    m.load_state_dict(ref.state_dict())
    m.cuda().eval(); ref.eval()
    h_a, h_t, m_a, m_t = _rand_batch(B, Ta, Tt, d, 5)
    with torch.no_grad():
        lr, br, zr = ref(h_a, h_t, m_a, m_t)
        lg, bg, zg = m(cu(h_a), cu(h_t), cu(m_a), cu(m_t))
    close(lg, lr, what="logits"); close(bg, br, what="beta"); close(zg, zr, what="z")


def test_text_longer_than_audio_raises_like_the_reference(H):
    """The gate aligns both streams to the text length (beta_gate_tacfn.py:98-116): with L_t > L_a the reference fails
    with a RuntimeError (shape mismatch); the drop-in refuses the same input with the same exception type."""
    kw = dict(d_model=128, num_emotions=2, n_heads=8, dropout=0.0, num_layers_fusion=1, num_layers_decoder=1)
    ref, m = O.FusionWithEmotionDecoder(**kw).eval(), H.FusionWithEmotionDecoder(**kw).cuda().eval()
    h_a, h_t, _, _ = _rand_batch(2, 7, 33, 128, 3, ragged=False)
    with torch.no_grad():
        with pytest.raises(RuntimeError):
            ref(h_a, h_t)
        with pytest.raises(RuntimeError):
            m(cu(h_a), cu(h_t))


def test_half_precision_parameters_are_refused_loudly(H):
    """model.bfloat16() would hand bf16 biases to kernels that read fp32: refused with a TypeError instead."""
    m = H.FusionWithEmotionDecoder(d_model=128, num_emotions=2, n_heads=8).cuda().bfloat16().eval()
    h_a, h_t, _, _ = _rand_batch(2, 8, 4, 128, 3, ragged=False)
    with torch.no_grad(), pytest.raises(TypeError):
        m(cu(h_a), cu(h_t))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_low_precision_inputs_keep_their_dtype(H, dtype):
    """bf16 / fp16 embeddings (what an autocast region hands over): z comes back in that dtype and everything agrees with the
    fp32-input result up to the rounding of the inputs."""
    torch.manual_seed(5)
    m = fusion(H, 128, 4).eval()
    h_a, h_t, m_a, m_t = _rand_batch(4, 32, 16, 128, 11)
    with torch.no_grad():
        l32, b32, z32 = m(cu(h_a), cu(h_t), cu(m_a), cu(m_t))
        lx, bx, zx = m(cu(h_a).to(dtype), cu(h_t).to(dtype), cu(m_a), cu(m_t))
    # z follows the embeddings' dtype; the per-emotion head multiplies z by fp32 parameters, so logits are fp32 (type
    # promotion does the same in the reference under autocast: emotion_decoder.py:150-158)
    assert zx.dtype == dtype and lx.dtype == torch.float32
    close(lx, l32, tol=3e-2, what="logits"); close(zx, z32, tol=3e-2, what="z"); close(bx, b32, tol=3e-2, what="beta")


@pytest.mark.parametrize("B,Ta,Tt,d,ne", [(3, 100, 40, 256, 5), (2, 130, 48, 768, 6)])
def test_train_step_with_dropout_equals_the_oracle_under_the_same_masks(H, monkeypatch, B, Ta, Tt, d, ne):
    """Train mode, dropout 0.1, whole model fwd+bwd, EXACT: the HIP path records every dropout site of its forward (seed, site,
    shape: _ops.DROP_LOG); tests/hashrng.py -- the host replica of the kernels' counter hash, itself pinned against the kernels
    in test_gpu_kernels.py -- rebuilds the keep-masks; the fp32 oracle then runs the same step with exactly those masks in
    place of torch's dropout (both paths visit their dropout sites in the reference's order: cross_modal_block_tacfn.py:70-125,
    emotion_decoder.py:42-59).  Loss, logits and every parameter gradient must agree to the bounds of the dropout-free tests:
    not statistics, the same function."""
    import numpy as np
    import hashrng
    from hri_emo_amd import _ops
    torch.manual_seed(1234)
    kw = dict(d_model=d, num_emotions=ne, n_heads=8, dropout=0.1)
    ref = O.FusionWithEmotionDecoder(**kw).train()
    m = H.FusionWithEmotionDecoder(**kw)
    m.load_state_dict(ref.state_dict())
    m.cuda().train()
    h_a, h_t, m_a, m_t = _rand_batch(B, Ta, Tt, d, 11)
    y = (torch.rand(B, ne, generator=torch.Generator().manual_seed(12)) < 0.3).float()
    word = int(_ops.seed_word(torch.device("cuda", 0)).item()) & ((1 << 64) - 1)
    log = []
    monkeypatch.setattr(_ops, "DROP_LOG", log)
    torch.manual_seed(77)
    loss_m, logits_m, ga_m, gt_m, gm = _train_step(m, cu(h_a), cu(h_t), cu(m_a), cu(m_t), cu(y))
    monkeypatch.setattr(_ops, "DROP_LOG", None)
    n_attn, n_rows = sum(e[0] == "attn" for e in log), sum(e[0] == "rows" for e in log)
    assert n_attn == 2 * 4 + 2 * 2 and n_rows == 2 * 6 + 2 * 4, (n_attn, n_rows)       # 2 fusion blocks, 2 decoder layers

    def keep_of(e, shape):
        seed = (e[1] + word) & ((1 << 64) - 1)
        if e[0] == "attn":
            _, _, site, B_, H_, Lq, Lk, p, b_off = e
            k = hashrng.attn_mask(seed, site, B_, H_, Lq, Lk, p, b_off)
        else:
            _, _, site, M, N, p, row_off = e
            k = hashrng.rows_mask(seed, site, M, N, p, row_off)
        assert int(np.prod(k.shape)) == int(np.prod(shape)), (e, tuple(shape))
        return torch.from_numpy(k.reshape(tuple(shape))), hashrng.inv_keep(e[-2] if e[0] == "attn" else e[5])

    cursor = [0]

    def replay_dropout(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        e = log[cursor[0]]
        cursor[0] += 1
        keep, scale = keep_of(e, x.shape)
        return x * (keep.to(x.dtype) * scale)

    monkeypatch.setattr(torch.nn.functional, "dropout", replay_dropout)
    loss_r, logits_r, ga_r, gt_r, gr = _train_step(ref, h_a, h_t, m_a, m_t, y)
    assert cursor[0] == len(log)                       # the oracle visited exactly the sites the HIP path logged, in order
    cursor[0] = 0
    _, _, ga_y, gt_y, gy = _train_step(ref, h_a, h_t, m_a, m_t, y, autocast_cpu=True)
    monkeypatch.undo()
    drop_rate = 1.0 - float(np.mean([keep_of(e, (e[3] * e[4] * e[5] * e[6],) if e[0] == "attn" else (e[3] * e[4],))[0].float().mean()
                                      for e in log]))
    assert abs(drop_rate - 0.1) < 5e-3, drop_rate
    close(loss_m.reshape(1), loss_r.reshape(1), what="loss"); close(logits_m, logits_r, what="logits")
    # The gate's first Linear: its gradient is a cancellation-heavy sum over only B = 2-3 pooled rows (dL/dw = sum over L x d of
    # dH . (A_n - T_n), then through sigmoid' and ReLU'); both bf16 paths -- this one and the reference under CPU autocast -- land
    # anywhere between 2 % and 11 % of the fp32 oracle there depending on the draw (scripts_dev/diag_dropout_exact.py, dropout on
    # or off), so the yardstick of one draw does not bound the other path's.  Every other parameter keeps the usual bound.
    gate = {"beta_gate.mlp.0.weight": 0.15, "beta_gate.mlp.0.bias": 0.15}
    assert_per_parameter_grads(gm, gy, gr, exceptions=gate, what=f"dropout-exact {B}x{Ta}x{Tt}x{d}")
    assert _rel(ga_m, ga_r) <= max(3e-2, 1.5 * _rel(ga_y, ga_r)), "d loss / d h_a"
    assert _rel(gt_m, gt_r) <= max(3e-2, 1.5 * _rel(gt_y, gt_r)), "d loss / d h_t"
    # and the masks matter: the oracle with torch's own dropout draws gives another loss
    torch.manual_seed(78)
    loss_other = _train_step(ref, h_a, h_t, m_a, m_t, y)[0]
    assert abs(float(loss_other) - float(loss_r)) > 1e-5


def test_train_mode_dropout_statistics(H):
    """dropout=0.1 train-mode forward: finite, differs from eval, and stays near it on average."""
    torch.manual_seed(0)
    m = fusion(H, 128, 4, p=0.1)
    h_a, h_t, m_a, m_t = _rand_batch(64, 32, 16, 128, 9)
    args = (cu(h_a), cu(h_t), cu(m_a), cu(m_t))
    with torch.no_grad():
        le, _, _ = m.eval()(*args)
        lt1, _, _ = m.train()(*args)
        lt2, _, _ = m.train()(*args)
    assert torch.isfinite(lt1).all() and not torch.equal(lt1, le) and not torch.equal(lt1, lt2)
    assert (lt1 - le).abs().mean().item() < 0.5 * le.abs().mean().item() + 0.1


def test_full_size_cfg2_properties(H):
    """BASELINE cfg 2 (d=768, T_a=400, T_t=128, N_e=6, B=64): utterances are independent, so
    (a) permuting the batch permutes the outputs bit-exactly, (b) a shard computed alone with its
    batch offset equals the same rows of the full batch bit-exactly -- in TRAIN mode, dropout on
    (the masks are keyed on the global utterance index), (c) grads are finite and non-zero."""
    torch.manual_seed(1234)
    m = H.FusionWithEmotionDecoder(d_model=768, num_emotions=6, n_heads=8, dropout=0.1).cuda()
    B = 64
    h_a, h_t, m_a, m_t = _rand_batch(B, 400, 128, 768, 77)
    h_a, h_t, m_a, m_t = h_a.cuda().bfloat16(), h_t.cuda().bfloat16(), m_a.cuda(), m_t.cuda()
    m.eval()
    with torch.no_grad():
        l0, b0, z0 = m(h_a, h_t, m_a, m_t)
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
        l1, b1, z1 = m(h_a[perm], h_t[perm], m_a[perm], m_t[perm])
    assert torch.equal(l1, l0[perm]) and torch.equal(z1, z0[perm]) and torch.equal(b1, b0[perm])
    m.train()
    with torch.no_grad():
        torch.manual_seed(42); lf, bf, zf = m(h_a, h_t, m_a, m_t)
        m.set_batch_offset(32)
        torch.manual_seed(42); ls, bs, zs = m(h_a[32:], h_t[32:], m_a[32:], m_t[32:])
        m.set_batch_offset(0)
    assert torch.equal(ls, lf[32:]) and torch.equal(zs, zf[32:]) and torch.equal(bs, bf[32:])
    y = (torch.rand(B, 6, device="cuda") < 0.3).float()
    logits, beta, z = m(h_a, h_t, m_a, m_t)
    O.train_step_loss(logits, beta, y).backward()
    for n, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0, n


def test_fused_weight_grad_accumulation_matches_autograd_path(H):
    """Parameter gradients written straight into pre-existing .grad buffers (_ops.GradSink: GEMM/column-
    reduce accumulate) equal the ones autograd accumulates from returned tensors, and really accumulate."""
    from hri_emo_amd.dp import GradBuckets
    torch.manual_seed(5)
    m = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=8, dropout=0.0).cuda().train()
    h_a, h_t, m_a, m_t = _rand_batch(6, 40, 24, 128, 21)
    args = (cu(h_a), cu(h_t), cu(m_a), cu(m_t))
    y = (torch.rand(6, 4, device="cuda") < 0.3).float()

    def step():
        logits, beta, _ = m(*args)
        O.train_step_loss(logits, beta, y).backward()

    step()                                                   # p.grad is None -> tensors returned to autograd
    ref = {n: p.grad.clone() for n, p in m.named_parameters()}
    buckets = GradBuckets(m.parameters())                    # flat fp32 buffer, p.grad are views into it
    buckets.zero_grad()
    step()                                                   # fused: kernels accumulate in place
    for n, p in m.named_parameters():
        assert p.grad.data_ptr() >= buckets.flat.data_ptr(), n
        # same partial sums, but the launch-boundary reduce adds them in another (fixed) order than the in-place one
        err = (p.grad - ref[n]).abs().max().item()
        assert err <= 1e-6 * max(1.0, ref[n].abs().max().item()), (n, err)
    step()                                                   # no zeroing: must now hold twice the gradient
    for n, p in m.named_parameters():
        err = (p.grad - 2 * ref[n]).abs().max().item()
        assert err <= 1e-5 * max(1.0, ref[n].abs().max().item()), (n, err)


def test_graph_captured_step_matches_eager_and_rccl_single_rank(H):
    """DataParallelStep.capture(): a hipGraph replay of zero-grad+fwd+loss+bwd gives the eager gradients
    (dropout off), fresh dropout masks per replay (dropout on), and the RCCL all-reduce path runs
    (world size 1 on this one-GPU box; multi-rank semantics are covered by tests/test_dp_gloo.py)."""
    import os
    import torch.distributed as dist
    from hri_emo_amd.dp import DataParallelStep
    from hri_emo_amd.train import fusion_step_loss
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29631")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        t = torch.ones(1 << 20, device="cuda")
        dist.all_reduce(t)
        assert float(t.sum()) == float(1 << 20)
        torch.manual_seed(3)
        m = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=8, dropout=0.0).cuda().train()
        h_a, h_t, m_a, m_t = _rand_batch(8, 48, 24, 128, 31)
        batch = (cu(h_a).bfloat16(), cu(h_t).bfloat16(), cu(m_a), cu(m_t), (torch.rand(8, 4, device="cuda") < 0.3).float())
        dp = DataParallelStep(m, fusion_step_loss, overlap=False)
        dp.set_global_batch(8)
        loss_e = dp.step(*batch).clone()
        g_e = dp.buckets.flat.clone()
        dp.capture(*batch)
        loss_g = dp.step(*batch)
        assert torch.equal(loss_g, loss_e) and torch.equal(dp.buckets.flat, g_e)
        dp.step(*batch)                                     # replay again: still the same (no accumulation leak)
        assert torch.equal(dp.buckets.flat, g_e)
        # new data through the static buffers
        h_a2, h_t2, m_a2, m_t2 = _rand_batch(8, 48, 24, 128, 32)
        dp.step(cu(h_a2).bfloat16(), cu(h_t2).bfloat16(), cu(m_a2), cu(m_t2), batch[4])
        assert not torch.equal(dp.buckets.flat, g_e)
        # dropout on: every replay must draw a different mask (device seed word bumped inside the graph)
        m2 = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=8, dropout=0.2).cuda().train()
        dp2 = DataParallelStep(m2, fusion_step_loss, overlap=False)
        dp2.step(*batch)
        dp2.capture(*batch)
        l1 = dp2.step(*batch).clone()
        l2 = dp2.step(*batch).clone()
        assert torch.isfinite(l1) and torch.isfinite(l2) and not torch.equal(l1, l2)
    finally:
        dist.destroy_process_group()


def test_mosei_wrapper_vs_golden_and_amp_gradscaler(H):
    """SURVEY 8(f) rank 1+2: MoseiFusionWithEmotionDecoder (d_audio=74, d_text=300 -> odd-K projection GEMMs,
    reference MOSEI defaults d_model=256 / 4 heads / hd=64) against the golden from the reference; and the
    MOSEI trainer's AMP pattern (train_mosei_fusion_seq_level_decoder.py:380-402): autocast + GradScaler
    scale/unscale/clip/step run unchanged on top of the custom autograd Functions."""
    g = load_golden("mosei_eval_train")
    ref = O.closed_form_init_(O.MoseiFusionWithEmotionDecoder(d_audio=74, d_text=300))
    m = H.MoseiFusionWithEmotionDecoder(d_audio=74, d_text=300)
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
    m.load_state_dict(ref.state_dict(), strict=True)
    m.cuda().eval()
    args = (cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]))
    with torch.no_grad():
        logits, beta, z = m(*args)
        l4, b4, z4, pack = m(*args, return_attention=True)
    close(logits, g["logits"], what="logits"); close(beta, g["beta"], what="beta"); close(z, g["z"], what="z")
    assert len(pack["encoder"]) == 2 and pack["decoder"][0].shape == (3, 6, 20)
    # gradients of the projections (dropout 0) vs the golden
    mt = H.MoseiFusionWithEmotionDecoder(d_audio=74, d_text=300, dropout=0.0)
    mt.load_state_dict(ref.state_dict())
    mt.cuda().train()
    l2, b2, _ = mt(*args)
    loss = O.train_step_loss(l2, b2, cu(g["y"]))
    loss.backward()
    close(loss.reshape(1), g["loss"], what="loss")
    # yardstick rule (assert_per_parameter_grads): the oracle's own CPU-autocast(bf16) error on the same parameter
    refy = O.closed_form_init_(O.MoseiFusionWithEmotionDecoder(d_audio=74, d_text=300, dropout=0.0)).train()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ly, by, _ = refy(g["h_a"], g["h_t"], g["mask_a"], g["mask_t"])
    O.train_step_loss(ly.float(), by.float(), g["y"]).backward()
    gyard = {n: p.grad for n, p in refy.named_parameters()}
    for name, key in (("audio_proj.weight", "g_audio_proj_w"), ("audio_proj.bias", "g_audio_proj_b"),
                      ("text_proj.weight", "g_text_proj_w"), ("text_proj.bias", "g_text_proj_b")):
        got = dict(mt.named_parameters())[name].grad
        em, ey = _rel(got, g[key]), _rel(gyard[name], g[key])
        assert em <= max(GRAD_FLOOR, GRAD_FACTOR * ey), (name, em, ey)
    # AMP + GradScaler step
    opt = torch.optim.AdamW(mt.parameters(), lr=1e-4)
    scaler = torch.amp.GradScaler("cuda")
    before = mt.audio_proj.weight.detach().clone()
    opt.zero_grad(set_to_none=True)
    with torch.amp.autocast("cuda"):
        l3, b3, _ = mt(*args)
        loss3 = O.train_step_loss(l3.float(), b3.float(), cu(g["y"]))
    scaler.scale(loss3).backward()
    scaler.unscale_(opt)
    torch.nn.utils.clip_grad_norm_(mt.parameters(), 5.0)
    scaler.step(opt)
    scaler.update()
    assert torch.isfinite(loss3) and not torch.equal(mt.audio_proj.weight.detach(), before)
    close(loss3.reshape(1), g["loss"], what="amp loss")


def test_legacy_block_and_fusion_classifier_vs_golden(H):
    """the legacy cross-modal block (what the reference's tests/test_cross_modal_block.py
    runs) and FusionClassifier (tests/test_fusion_classifier.py), sequence- and utterance-level inputs."""
    from hri_emo_amd.models.cross_modal_block import CrossModalTransformer as LegacyCMT
    g = load_golden("legacy_eval")
    ref = O.closed_form_init_(O.LegacyCrossModalTransformer(2, 128, 8, 0.1))
    leg = LegacyCMT(num_layers=2, d_model=128, n_heads=8, dropout=0.1)
    leg.load_state_dict(ref.state_dict(), strict=True)
    leg.cuda().eval()
    with torch.no_grad():
        oa, ot = leg(cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]))
    close(oa, g["leg_a"], what="legacy a"); close(ot, g["leg_t"], what="legacy t")
    refc = O.closed_form_init_(O.FusionClassifier(128, 4, 8, 2, 32))
    clf = H.FusionClassifier(d_model=128, num_classes=4, n_heads=8, num_layers=2, beta_hidden=32)
    clf.load_state_dict(refc.state_dict(), strict=True)
    clf.cuda().eval()
    with torch.no_grad():
        logits, beta, pooled = clf(cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]))
        l2, b2, p2 = clf(cu(g["u_a"]), cu(g["u_t"]))
    close(logits, g["clf_logits"], 1e-2, "clf logits"); close(beta, g["clf_beta"], what="clf beta")
    close(pooled, g["clf_pooled"], 1e-2, "clf pooled")
    close(l2, g["u_logits"], 1e-2, "utt logits"); close(p2, g["u_pooled"], 1e-2, "utt pooled")
    # trains: gradients reach the head and the fusion blocks
    clf.train()
    out, _, _ = clf(cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]))
    out.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in clf.parameters())


def test_legacy_scalar_beta_gate_vs_golden(H):
    """models/beta_gate.py (scalar gate, no LayerNorm) -- what the reference's tests/test_beta_gate.py
    builds: ragged masks with unequal lengths, gradients w.r.t. inputs and the MLP, and the test's own [B,1,d] call."""
    from hri_emo_amd.models.beta_gate import BetaGate as LegacyGate
    g = load_golden("legacy_gate")
    ref = O.closed_form_init_(O.LegacyBetaGate(128, 32))
    gate = LegacyGate(d_model=128, hidden_dim=32)
    gate.load_state_dict(ref.state_dict(), strict=True)
    gate.cuda()
    h_a, h_t = cu(g["h_a"]).requires_grad_(True), cu(g["h_t"]).requires_grad_(True)
    hf, beta = gate(h_a, h_t, cu(g["mask_a"]), cu(g["mask_t"]))
    close(hf, g["h_fusion"], 1e-2, "legacy gate h_fusion (bf16 output)"); close(beta, g["beta"], what="legacy gate beta")
    ((hf * cu(g["c"])).sum() + 3.0 * beta.sum()).backward()
    for got, key in [(h_a.grad, "g_h_a"), (h_t.grad, "g_h_t"), (gate.mlp[0].weight.grad, "g_w1"), (gate.mlp[0].bias.grad, "g_b1"),
                     (gate.mlp[2].weight.grad, "g_w2"), (gate.mlp[2].bias.grad, "g_b2")]:
        assert _rel(got, g[key]) < 2e-2, (key, _rel(got, g[key]))
    with torch.no_grad():
        hu, bu = gate(cu(g["u_a"]), cu(g["u_t"]))
    close(hu, g["u_h"], 1e-2, "utterance-level h"); close(bu, g["u_beta"], what="utterance-level beta")


def test_full_trainer_step_clip_adamw_vs_golden(H):
    """The trainer's whole step on the golden inputs (train_fusion_seq_level_decoder.py:310-334): fwd -> loss ->
    backward -> clip_grad_norm_(5.0) -> AdamW(lr 1e-4, wd 1e-2).step(): total gradient norm and the per-parameter
    post-step deltas against the values recorded from the reference (harness row)."""
    g = load_golden("cfg1_train_p0")
    ref = O.closed_form_init_(O.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=8, dropout=0.0))
    m = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=8, dropout=0.0)
    m.load_state_dict(ref.state_dict())
    m.cuda().train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-2)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    logits, beta, _ = m(cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]))
    loss = O.train_step_loss(logits, beta, cu(g["y"]))
    opt.zero_grad()
    loss.backward()
    tn = torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
    opt.step()
    close(loss.reshape(1), g["loss"], what="loss")
    assert abs(float(tn) - float(g["total_grad_norm"])) <= 2e-2 * float(g["total_grad_norm"]), (float(tn), float(g["total_grad_norm"]))
    errs = []
    for n, p in m.named_parameters():
        dn, rn = float((p.detach() - before[n]).norm()), float(g["delta.norm." + n])
        errs.append((abs(dn - rn) / max(rn, 1e-12), n))
    errs.sort()
    assert errs[len(errs) // 2][0] < 2e-2 and errs[len(errs) * 9 // 10][0] < 1e-1, errs[-4:]


def test_fused_clip_adamw_matches_torch(H):
    """hri_emo_amd.optim.FusedClipAdamW (two HIP passes over the flat gradient / parameter / moment buffers) against
    torch.nn.utils.clip_grad_norm_(5.0) + torch.optim.AdamW(lr 1e-4, wd 1e-2) over three trainer steps of the
    small fusion model, same gradients on both sides (train_fusion_seq_level_decoder.py:332-334)."""
    from hri_emo_amd.dp import GradBuckets
    from hri_emo_amd.optim import FusedClipAdamW
    g = load_golden("cfg1_train_p0")
    torch.manual_seed(7)
    m = fusion(H, 128, 4, p=0.0).train()
    twin = {n: p.detach().clone().requires_grad_(True) for n, p in m.named_parameters()}
    ref_opt = torch.optim.AdamW(list(twin.values()), lr=1e-4, weight_decay=1e-2)
    buckets = GradBuckets(m.parameters(), overlap=False)
    opt = FusedClipAdamW(buckets, lr=1e-4, weight_decay=1e-2, max_norm=5.0)
    for step in range(3):
        buckets.zero_grad()
        logits, beta, _ = m(cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]))
        (O.train_step_loss(logits, beta, cu(g["y"])) * (40.0 if step == 1 else 1.0)).backward()     # step 1 clips
        for n, p in m.named_parameters():
            twin[n].grad = p.grad.detach().clone()
        tn_ref = torch.nn.utils.clip_grad_norm_(list(twin.values()), 5.0)
        ref_opt.step()
        tn = opt.step()
        assert abs(float(tn) - float(tn_ref)) <= 1e-5 * float(tn_ref), (step, float(tn), float(tn_ref))
        if step == 1:
            assert float(tn_ref) > 5.0, "the scaled step is meant to exercise clipping"
        for n, p in m.named_parameters():
            err = (p.detach() - twin[n].detach()).abs().max().item()
            assert err <= 2e-6 * max(1.0, twin[n].detach().abs().max().item()), (step, n, err)
    # the model still runs on the re-homed parameter storage
    with torch.no_grad():
        m.eval()(cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]))


def test_eager_steps_with_fused_optimizer_follow_torch_adamw(H):
    """Eager (no hipGraph) DataParallelStep + FusedClipAdamW for several steps against the same model driven by
    clip_grad_norm_ + torch.optim.AdamW: the fused optimizer rewrites the flat parameter buffer through raw pointers, so the
    bf16 weight shadows must be told (WEIGHTS_EPOCH) -- otherwise every forward after the first runs on the initial weights.
    lr is large so a stale-weights forward shows up in the very next loss."""
    from hri_emo_amd.dp import DataParallelStep
    from hri_emo_amd.optim import FusedClipAdamW
    from hri_emo_amd.train import fusion_step_loss
    g = load_golden("cfg1_train_p0")
    batch = (cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]), cu(g["y"]))
    torch.manual_seed(9)
    m1 = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=8, dropout=0.0).cuda().train()
    m2 = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=8, dropout=0.0)
    m2.load_state_dict(m1.state_dict())
    m2.cuda().train()
    dp = DataParallelStep(m1, fusion_step_loss, overlap=False)
    opt1 = FusedClipAdamW(dp.buckets, lr=1e-3, weight_decay=1e-2, max_norm=5.0)
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-3, weight_decay=1e-2)
    l1s, l2s = [], []
    for step in range(4):
        l1s.append(float(dp.step(*batch)))                       # eager: zero-grad + fwd + loss + bwd
        opt1.step()
        opt2.zero_grad(set_to_none=True)
        logits, beta, _ = m2(*batch[:4])
        loss2 = fusion_step_loss(logits, beta, batch[4])
        loss2.backward()
        torch.nn.utils.clip_grad_norm_(m2.parameters(), 5.0)
        opt2.step()
        l2s.append(float(loss2))
    assert abs(l1s[1] - l1s[0]) > 1e-3 and abs(l1s[3] - l1s[2]) > 1e-4, ("the updates must be visible in the loss", l1s)
    for k, (a, b) in enumerate(zip(l1s, l2s)):
        # both sides run the same bf16 kernels; they differ by the order of fp32 gradient sums, amplified step by step
        assert abs(a - b) <= 1e-2 * max(1.0, abs(b)), (l1s, l2s)
        if k:   # and by far less than one update moves the loss (a forward on stale weights would repeat the previous loss)
            assert abs(a - b) < 0.25 * abs(l2s[k] - l2s[k - 1]), (l1s, l2s)
    with torch.no_grad():
        o1, o2 = m1.eval()(*batch[:4]), m2.eval()(*batch[:4])
    close(o1[0], o2[0], 2e-2, "logits after 4 steps")


def test_shared_activation_gradients_joined_in_the_gemm_epilogue(H, monkeypatch):
    """_ops.GradJoin: the two consumers of each self-attention output (and the decoder layers' shared memory) hand their gradients
    to ONE dX GEMM instead of an autograd add launch.  Same gradients as with autograd summing them (HRIEMO_GRAD_JOIN=0 path), to
    one bf16 rounding of the summed activation gradient; and a join whose partner never ran must raise, not lose a gradient."""
    from hri_emo_amd import _ops
    from hri_emo_amd.train import fusion_step_loss
    torch.manual_seed(5)
    kw = dict(d_model=256, num_emotions=5, n_heads=8, dropout=0.0)
    m = H.FusionWithEmotionDecoder(**kw).cuda().train()
    h_a, h_t, m_a, m_t = _rand_batch(4, 90, 36, 256, 23)
    y = (torch.rand(4, 5, generator=torch.Generator().manual_seed(2)) < 0.3).float().cuda()
    grads = []
    for on in (True, False):
        monkeypatch.setattr(_ops, "GRAD_JOIN", on)
        m.zero_grad(set_to_none=True)
        logits, beta, z = m(cu(h_a), cu(h_t), cu(m_a), cu(m_t))
        fusion_step_loss(logits, beta, y).backward()
        grads.append({n: p.grad.detach().float().clone() for n, p in m.named_parameters()})
    worst = max(float((grads[0][n] - grads[1][n]).norm() / grads[1][n].norm().clamp_min(1e-20)) for n in grads[0])
    assert worst <= 1e-2, worst              # the joined path rounds (g1 + g2) once, autograd rounds g1, g2 and their sum
    monkeypatch.setattr(_ops, "GRAD_JOIN", True)
    # a stand-alone block trained on ONE of its outputs: the text branch's consumers never run, so no join may be active there
    blk = H.CrossModalBlock(256, 8, 0.0).cuda().train()
    oa, ot = blk(cu(h_a), cu(h_t), cu(m_a), cu(m_t))
    oa.float().pow(2).mean().backward()
    assert all(torch.isfinite(p.grad).all() for p in blk.parameters() if p.grad is not None)
    j = _ops.GradJoin(2)
    x = torch.randn(2, 8, 128, device="cuda", requires_grad=True)
    w = torch.nn.Parameter(torch.randn(384, 128, device="cuda") * 0.05)
    b = torch.nn.Parameter(torch.zeros(384, device="cuda"))
    kv = _ops.KVProjFn.apply(x, w, b, _ops.Shadows(), j)
    with pytest.raises(RuntimeError, match="GradJoin"):
        kv.float().sum().backward()          # the join's other consumer never arrives


@pytest.mark.parametrize("two_streams", [True, False])
def test_shared_input_projection_equals_separate_projections(H, monkeypatch, two_streams):
    """_ops.SharedProjFn: ONE N = 3d GEMM per self-attention output ([Q of its own cross-attention | K, V of the other one],
    models/cross_modal_block_tacfn.py:98-104,111-117) instead of a Q and a K|V launch, one K = 3d dX GEMM in backward.  Same
    outputs (bit for bit: the GEMM tiles do not depend on N) and the same gradients (one bf16 rounding of the summed activation
    gradient apart) as the separate projections (HRIEMO_SHARED_PROJ=0); the state_dict is untouched; also a stand-alone block
    trained on one output only (one of the two gradient halves never arrives)."""
    from hri_emo_amd import _ops
    from hri_emo_amd.train import fusion_step_loss
    monkeypatch.setattr(_ops, "TWO_STREAMS", two_streams)
    torch.manual_seed(11)
    m = H.FusionWithEmotionDecoder(d_model=256, num_emotions=5, n_heads=8, dropout=0.0).cuda().train()
    keys = list(m.state_dict().keys())
    h_a, h_t, m_a, m_t = _rand_batch(4, 90, 36, 256, 31)
    y = (torch.rand(4, 5, generator=torch.Generator().manual_seed(4)) < 0.3).float().cuda()
    outs, grads = [], []
    for on in (True, False):
        monkeypatch.setattr(_ops, "SHARED_PROJ", on)
        m.zero_grad(set_to_none=True)
        logits, beta, z = m(cu(h_a), cu(h_t), cu(m_a), cu(m_t))
        fusion_step_loss(logits, beta, y).backward()
        outs.append((logits.detach().clone(), beta.detach().clone(), z.detach().clone()))
        grads.append({n: p.grad.detach().float().clone() for n, p in m.named_parameters()})
    assert list(m.state_dict().keys()) == keys
    for a_, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a_, b_)
    worst = max(float((grads[0][n] - grads[1][n]).norm() / grads[1][n].norm().clamp_min(1e-20)) for n in grads[0])
    assert worst <= 1e-2, worst
    monkeypatch.setattr(_ops, "SHARED_PROJ", True)
    blk = H.CrossModalBlock(256, 8, 0.0).cuda().train()
    ref = H.CrossModalBlock(256, 8, 0.0).cuda().train()
    ref.load_state_dict(blk.state_dict())
    oa, ot = blk(cu(h_a), cu(h_t), cu(m_a), cu(m_t))
    oa.float().pow(2).mean().backward()
    monkeypatch.setattr(_ops, "SHARED_PROJ", False)
    ra, rt = ref(cu(h_a), cu(h_t), cu(m_a), cu(m_t))
    ra.float().pow(2).mean().backward()
    for (n, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        assert (p.grad is None) == (q.grad is None), n
        if p.grad is not None:
            assert float((p.grad - q.grad).norm()) <= 1e-2 * max(float(q.grad.norm()), 1e-20), n


def test_partial_fine_tuning_with_a_frozen_cross_attention(H, monkeypatch):
    """a join is only created where both consumers of a shared activation get a backward node.  Layer 0 frozen except
    its audio->text cross-attention, inputs without gradient: the t2a K|V projection node never exists, and the a2t core must
    neither deposit into a join nor raise.  Same for a decoder whose memory needs no gradient and whose second layer's
    cross-attention in-projection is frozen.  Gradients equal the HRIEMO_GRAD_JOIN=0 path."""
    from hri_emo_amd import _ops
    from hri_emo_amd.train import fusion_step_loss
    torch.manual_seed(6)
    m = H.FusionWithEmotionDecoder(d_model=256, num_emotions=5, n_heads=8, dropout=0.0).cuda().train()
    l0 = m.cross_modal.layers[0]
    for n, p in l0.named_parameters():
        p.requires_grad_(n.startswith("attn_a2t.") or n.startswith("norm_a1."))
    h_a, h_t, m_a, m_t = _rand_batch(4, 90, 36, 256, 29)
    y = (torch.rand(4, 5, generator=torch.Generator().manual_seed(3)) < 0.3).float().cuda()
    grads = []
    for on in (True, False):
        monkeypatch.setattr(_ops, "GRAD_JOIN", on)
        m.zero_grad(set_to_none=True)
        logits, beta, z = m(cu(h_a), cu(h_t), cu(m_a), cu(m_t))
        fusion_step_loss(logits, beta, y).backward()
        grads.append({n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None})
    assert set(grads[0]) == set(grads[1]) and "cross_modal.layers.0.attn_a2t.in_proj_weight" in grads[0]
    assert not any(n.startswith("cross_modal.layers.0.attn_t2a") for n in grads[0])
    worst = max(float((grads[0][n] - grads[1][n]).norm() / grads[1][n].norm().clamp_min(1e-20)) for n in grads[0])
    assert worst <= 1e-2, worst
    monkeypatch.setattr(_ops, "GRAD_JOIN", True)
    dec = H.EmotionDecoder(d_model=256, num_emotions=5, n_heads=8, num_layers=2, dropout=0.0).cuda().train()
    dec.layers[1].cross_attn.in_proj_weight.requires_grad_(False)
    dec.layers[1].cross_attn.in_proj_bias.requires_grad_(False)
    mem = torch.randn(4, 36, 256, device="cuda")                    # no gradient wanted for the memory
    zq, lg = dec(mem, cu(m_t))
    lg.float().pow(2).mean().backward()
    assert dec.layers[0].cross_attn.in_proj_weight.grad is not None and dec.layers[1].cross_attn.in_proj_weight.grad is None
    assert torch.isfinite(dec.layers[0].cross_attn.in_proj_weight.grad).all()


def test_captured_step_replays_bit_identically_from_one_seed(H):
    """Every kernel sums in a fixed order: the captured training step (dropout 0.1, both streams, single-pass attention backward,
    deferred reduces and weight gradients) replayed from the SAME dropout seed word must reproduce loss and all gradient words
    bit for bit.  A race, a lost lane or an instruction hazard shows up here as differing words (scripts_dev/soak_step.py is
    the long version at the cfg-2 size)."""
    from hri_emo_amd import _ops
    from hri_emo_amd.dp import DataParallelStep
    from hri_emo_amd.train import fusion_step_loss
    torch.manual_seed(9)
    m = H.FusionWithEmotionDecoder(d_model=768, num_emotions=6, n_heads=8, dropout=0.1).cuda().train()
    h_a, h_t, m_a, m_t = _rand_batch(16, 400, 128, 768, 77)
    y = (torch.rand(16, 6, generator=torch.Generator().manual_seed(3)) < 0.3).float()
    batch = (cu(h_a).bfloat16(), cu(h_t).bfloat16(), cu(m_a), cu(m_t), cu(y))
    dp = DataParallelStep(m, fusion_step_loss, overlap=False)
    dp.set_global_batch(16)
    dp.step(*batch)
    dp.capture(*batch)
    sw = _ops.seed_word(batch[0].device)
    ref = None
    for i in range(12):
        sw.fill_(4242)
        loss = dp.step(*batch)
        torch.cuda.synchronize()
        g = dp.buckets.flat.view(torch.int32)
        if ref is None:
            ref, ref_loss = g.clone(), loss.clone()
            assert torch.isfinite(dp.buckets.flat).all() and float(dp.buckets.flat.norm()) > 0
            continue
        assert torch.equal(loss, ref_loss), (i, float(loss), float(ref_loss))
        assert torch.equal(g, ref), (i, int((g != ref).sum()))


def test_trimmed_batch_gives_the_same_outputs(H):
    """dropping the columns that are PAD for every sample (data.trim_padding) must not change what
    the model returns -- PAD keys are masked, PAD query rows feed nothing."""
    from hri_emo_amd import data
    torch.manual_seed(11)
    m = fusion(H, 128, 4, p=0.0).eval()
    B, Ta, Tt, d = 6, 80, 32, 128
    g = torch.Generator().manual_seed(12)
    h_a, h_t = torch.randn(B, Ta, d, generator=g), torch.randn(B, Tt, d, generator=g)
    va, vt = torch.randint(20, 50, (B,), generator=g), torch.randint(8, 20, (B,), generator=g)
    m_a, m_t = torch.arange(Ta)[None] >= va[:, None], torch.arange(Tt)[None] >= vt[:, None]
    ta, tma, tt, tmt = data.trim_padding(h_a, m_a, h_t, m_t)
    assert ta.shape[1] == int(va.max()) and tt.shape[1] == int(vt.max())
    with torch.no_grad():
        full = m(cu(h_a), cu(h_t), cu(m_a), cu(m_t))
        trim = m(cu(ta), cu(tt), cu(tma), cu(tmt))
    for a, b, what in zip(full, trim, ("logits", "beta", "z")):
        close(b, a.float().cpu(), what=what)


def _dp_overlap_worker(rank, world, port, q):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share GPU 0; gradients cross through the host
    try:
        import hri_emo_amd as H
        from hri_emo_amd.dp import DataParallelStep
        from hri_emo_amd.train import fusion_step_loss
        torch.cuda.set_device(0)
        torch.manual_seed(3)
        m = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=8, dropout=0.0).cuda().train()
        h_a, h_t, m_a, m_t = _rand_batch(8, 48, 24, 128, 31)
        y = (torch.rand(8, 4, generator=torch.Generator().manual_seed(5)) < 0.3).float()
        dp = DataParallelStep(m, fusion_step_loss, bucket_bytes=256 << 10, overlap=True)       # several buckets, hooks on
        lo, hi = dp.set_global_batch(8)
        for _ in range(2):
            dp.step(h_a[lo:hi].cuda().bfloat16(), h_t[lo:hi].cuda().bfloat16(), m_a[lo:hi].cuda(), m_t[lo:hi].cuda(), y[lo:hi].cuda())
        torch.cuda.synchronize()
        if rank == 0:
            q.put((dp.buckets.flat.cpu().numpy(), len(dp.buckets.buckets)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_overlapped_allreduce_equals_single_rank(H):
    """N>1 path as bench.py runs it (eager step, matrix-gradient buckets all-reduced from gradient-ready hooks while
    backward is still running, bias/LayerNorm bucket after the launch-boundary reduce), rehearsed with two ranks on
    ONE GPU over gloo: averaged gradients == the single-rank step on the concatenated batch."""
    import torch.multiprocessing as mp
    from hri_emo_amd.dp import DataParallelStep
    from hri_emo_amd.train import fusion_step_loss
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200
    procs = [ctx.Process(target=_dp_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    flat2, nb = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert nb > 2
    torch.manual_seed(3)
    m = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=8, dropout=0.0).cuda().train()
    h_a, h_t, m_a, m_t = _rand_batch(8, 48, 24, 128, 31)
    y = (torch.rand(8, 4, generator=torch.Generator().manual_seed(5)) < 0.3).float()
    dp = DataParallelStep(m, fusion_step_loss, bucket_bytes=256 << 10, overlap=False)
    dp.set_global_batch(8)
    dp.step(cu(h_a).bfloat16(), cu(h_t).bfloat16(), cu(m_a), cu(m_t), cu(y))
    ref = dp.buckets.flat.cpu()
    got = torch.from_numpy(flat2)
    assert got.shape == ref.shape
    per_bucket = [(i_, s_, e_, round(_rel(got[s_:e_], ref[s_:e_]), 5)) for i_, (s_, e_, _) in enumerate(dp.buckets.buckets)
                  if _rel(got[s_:e_], ref[s_:e_]) > 1e-4] + [len(dp.buckets.buckets)]
    assert _rel(got, ref) < 2e-3, (_rel(got, ref), per_bucket)
    for (s, e, _) in dp.buckets.buckets:                     # every bucket arrived, none twice
        assert _rel(got[s:e], ref[s:e]) < 5e-3, (s, e, _rel(got[s:e], ref[s:e]))


def test_bench_two_rank_control_flow_rehearsal():
    """bench.py for N>1, rehearsed with both ranks on this box's single GPU over gloo (HRIEMO_DIST_BACKEND): every leg must run
    to the one JSON line on rank 0 without a rank-local step launching a collective the other rank never joins.  Started the way
    the driver starts the N=1 run -- plain `python bench.py --gpus 2`, no launcher, no WORLD_SIZE -- so bench.py itself has to
    become the launcher (torch.distributed.run as a child process) and relay the ranks' output."""
    import json, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HRIEMO_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch-per-gpu", "8"]
    r = subprocess.run(cmd, env=env, cwd=repo, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["config"]["launch"] == "eager" and d["value"] > 0
    assert "roofline" in d and d["roofline"]["frac"] > 0 and d["cross_attention"] is not None


def _nccl_worker(rank, world, port, q):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        import hri_emo_amd as H
        from hri_emo_amd.dp import DataParallelStep
        from hri_emo_amd.train import fusion_step_loss
        torch.manual_seed(3)
        m = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=8, dropout=0.0).cuda().train()
        h_a, h_t, m_a, m_t = _rand_batch(8, 48, 24, 128, 31)
        y = (torch.rand(8, 4, generator=torch.Generator().manual_seed(5)) < 0.3).float()
        res = {}
        for name, kw in (("eager_overlap_fp32", dict(overlap=True)), ("eager_overlap_bf16", dict(overlap=True, comm_dtype=torch.bfloat16)),
                         ("replay_then_exchange", dict(overlap=False))):
            dp = DataParallelStep(m, fusion_step_loss, bucket_bytes=256 << 10, **kw)
            lo, hi = dp.set_global_batch(8)
            b = (h_a[lo:hi].cuda().bfloat16(), h_t[lo:hi].cuda().bfloat16(), m_a[lo:hi].cuda(), m_t[lo:hi].cuda(), y[lo:hi].cuda())
            dp.step(*b)
            if name == "replay_then_exchange":
                dp.capture(*b)
            for _ in range(2):
                dp.step(*b)
            torch.cuda.synchronize()
            res[name] = dp.buckets.flat.cpu().numpy()
        if rank == 0:
            q.put(res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the RCCL exchange over xGMI (one-GPU boxes rehearse it over gloo)")
def test_two_gpus_rccl_exchange_equals_single_rank(H):
    """The N>1 path on REAL RCCL the moment a multi-GPU box runs the suite: two ranks on two GPUs, eager step with the
    bucket all-reduces launched from gradient-ready hooks beside the two branch streams (fp32 and bf16 buckets), and the
    captured step with the exchange after the replay -- each must equal the single-rank step on the concatenated batch."""
    import torch.multiprocessing as mp
    from hri_emo_amd.dp import DataParallelStep
    from hri_emo_amd.train import fusion_step_loss
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 90
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    torch.manual_seed(3)
    m = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=8, dropout=0.0).cuda().train()
    h_a, h_t, m_a, m_t = _rand_batch(8, 48, 24, 128, 31)
    y = (torch.rand(8, 4, generator=torch.Generator().manual_seed(5)) < 0.3).float()
    dp = DataParallelStep(m, fusion_step_loss, bucket_bytes=256 << 10, overlap=False)
    dp.set_global_batch(8)
    dp.step(cu(h_a).bfloat16(), cu(h_t).bfloat16(), cu(m_a), cu(m_t), cu(y))
    ref = dp.buckets.flat.cpu()
    for name, flat in res.items():
        got = torch.from_numpy(flat)
        assert _rel(got, ref) < (1e-2 if "bf16" in name else 2e-3), (name, _rel(got, ref))


def test_device_prefetcher_overlapped_copies_hand_out_the_right_batches(H):
    """hri_emo_amd.data.DevicePrefetcher (the trainer's .to(device) at the top of every step,
    scripts/fusion/train_fusion_seq_level_decoder.py:306-308, as pinned double buffers on a copy stream): with ring slots being
    reused (12 batches through depth 2 and 3), pageable and pinned sources, each batch must arrive intact while the consumer
    stream is kept busy between hand-outs."""
    from hri_emo_amd.data import DevicePrefetcher
    g = torch.Generator().manual_seed(3)
    host = []
    for i in range(12):
        a_ = torch.randn(4, 50, 64, generator=g)
        host.append((a_.pin_memory() if i % 2 else a_, torch.randn(4, 20, 64, generator=g), torch.rand(4, 50) < 0.3, None, torch.full((4, 3), float(i))))
    busy = torch.randn(2048, 2048, device="cuda")
    for depth in (2, 3):
        n = 0
        for i, b in enumerate(DevicePrefetcher(host, "cuda", depth=depth, dtypes=(torch.bfloat16, torch.bfloat16, None, None, None))):
            assert b[3] is None and b[0].is_cuda and b[0].dtype == torch.bfloat16
            keep = [t.clone() for t in b if t is not None]
            for _ in range(3):
                busy = (busy @ busy).clamp(-1, 1)          # the "step": the next batch's copy runs beside it
            assert torch.equal(keep[0].cpu(), host[i][0].bfloat16()) and torch.equal(keep[1].cpu(), host[i][1].bfloat16())
            assert torch.equal(keep[2].cpu(), host[i][2]) and torch.equal(keep[3].cpu(), host[i][4])
            n += 1
        assert n == 12


_CAPTURED_EXCHANGE_SCRIPT = r"""
import os, sys, socket
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import hri_emo_amd as H
from hri_emo_amd.dp import DataParallelStep
from hri_emo_amd.train import fusion_step_loss
with socket.socket() as so:
    so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.manual_seed(21)
m = H.FusionWithEmotionDecoder(d_model=256, num_emotions=5, n_heads=8, dropout=0.0).cuda().train()
g = torch.Generator().manual_seed(41)
B, Ta, Tt, d = 6, 90, 36, 256
h_a, h_t = torch.randn(B, Ta, d, generator=g), torch.randn(B, Tt, d, generator=g)
la = torch.randint(Ta // 2, Ta + 1, (B,), generator=g); lt = torch.randint(Tt // 2, Tt + 1, (B,), generator=g)
y = (torch.rand(B, 5, generator=torch.Generator().manual_seed(5)) < 0.3).float().cuda()
batch = (h_a.cuda().bfloat16(), h_t.cuda().bfloat16(), (torch.arange(Ta)[None] >= la[:, None]).cuda(), (torch.arange(Tt)[None] >= lt[:, None]).cuda(), y)
dp = DataParallelStep(m, fusion_step_loss, bucket_bytes=512 << 10, overlap=True, force_exchange=True)
assert len(dp.buckets.buckets) > 3 and dp.buckets._hooks
dp.step(*batch)                                   # eager: hooks launch the collectives during backward
torch.cuda.synchronize()
ref = dp.buckets.flat.clone()
assert float(ref.norm()) > 0
dp.capture(*batch, collectives=True)
for _ in range(3):
    loss = dp.step(*batch)
    torch.cuda.synchronize()
    assert torch.equal(dp.buckets.flat, ref)
dp.use_graph(False)                               # and back to eager launches with the hook-driven exchange
dp.step(*batch)
torch.cuda.synchronize()
assert torch.equal(dp.buckets.flat, ref)
dp.buckets.close()
dist.destroy_process_group()
print("CAPTURED-EXCHANGE-OK")
"""


def _run_rccl_script(script, *args):
    """one-rank RCCL scripts run in their own process: the captured exchange can trip an upstream race in torch's NCCL watchdog
    thread (it polls a work whose end event was recorded during the capture: hipErrorCapturedEvent -> std::terminate, ~1 run in 8
    on ROCm 7.2 / torch 2.10), which must not take the suite down; that signature is an expected failure of the run"""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", script, repo, *args], cwd=repo, capture_output=True, text=True, timeout=420)
    if r.returncode != 0 and "hipErrorCapturedEvent" in r.stderr and "watchdog" in r.stderr:
        pytest.xfail("torch's NCCL watchdog queried an event recorded inside the capture (upstream race, ~1 run in 8)")
    return r


def test_gradient_exchange_captured_inside_the_step_graph():
    """dp.DataParallelStep.capture(collectives=True): the bucket all-reduces are launched from the gradient-ready hooks WHILE the
    backward is being captured and replayed with the step.  One rank over RCCL (a one-rank all-reduce is the
    identity, but the collective kernels, the process group's communication stream and its fork / join inside the capture are
    all real): the replayed gradients must equal the eager step's bit for bit, replay after replay, several buckets.  (The ORDER
    of the collectives against gradient production is the next test's subject.)"""
    r = _run_rccl_script(_CAPTURED_EXCHANGE_SCRIPT)
    assert r.returncode == 0 and "CAPTURED-EXCHANGE-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


_EXCHANGE_ORDER_SCRIPT = r"""
import os, sys, socket
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
import hri_emo_amd as H
from hri_emo_amd.dp import DataParallelStep
from hri_emo_amd.train import fusion_step_loss
mode = sys.argv[2]
with socket.socket() as so:
    so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.manual_seed(23)
m = H.FusionWithEmotionDecoder(d_model=256, num_emotions=5, n_heads=8, dropout=0.1).cuda().train()
names = [n for n, _ in m.named_parameters()]
batches = []
for sd in (51, 52, 53):
    g = torch.Generator().manual_seed(sd)
    B, Ta, Tt, d = 6, 90, 36, 256
    h_a, h_t = torch.randn(B, Ta, d, generator=g), torch.randn(B, Tt, d, generator=g)
    la = torch.randint(Ta // 2, Ta + 1, (B,), generator=g); lt = torch.randint(Tt // 2, Tt + 1, (B,), generator=g)
    y = (torch.rand(B, 5, generator=g) < 0.3).float()
    batches.append((h_a.cuda().bfloat16(), h_t.cuda().bfloat16(), (torch.arange(Ta)[None] >= la[:, None]).cuda(),
                    (torch.arange(Tt)[None] >= lt[:, None]).cuda(), y.cuda()))
dp = DataParallelStep(m, fusion_step_loss, bucket_bytes=2 << 20, overlap=True, force_exchange=True,
                      launch_from="main" if mode == "eager_main" else "notify")
assert len(dp.buckets.buckets) > 6 and dp.buckets._hooks
dp.buckets.enable_launch_snapshots()
dp.step(*batches[0])
if mode == "captured":
    dp.capture(*batches[0], collectives=True)
for it in range(12):
    dp.step(*batches[it % 3])
    torch.cuda.synchronize()
    assert float(dp.buckets.flat.norm()) > 0
    bad = dp.buckets.launch_snapshot_mismatches()
    assert not bad, (mode, it, [(b, n, names[o] if o is not None else None) for b, n, o in bad])
dp.buckets.flat[5].add_(1.0)                  # the hook itself must be able to fail
assert dp.buckets.launch_snapshot_mismatches()
dp.buckets.close()
dist.destroy_process_group()
print("EXCHANGE-ORDER-OK", mode, len(dp.buckets.buckets))
"""


@pytest.mark.parametrize("mode", ["eager_notify", "eager_main", "captured"])
def test_every_collective_is_launched_behind_the_gradients_of_its_bucket(mode):
    """the ORDER of the gradient exchange against gradient production, made testable on one GPU.  With
    GradBuckets.enable_launch_snapshots() every bucket is copied to a side buffer at the exact point its all-reduce is enqueued
    (on the launching stream, in front of the collective; inside the capture the copy is part of the graph).  One rank over RCCL:
    the all-reduce is the identity, so a collective that was launched before one of its gradients had been produced leaves the
    snapshot different from the final buffer.  Checked over 12 steps each for the eager step launching from the notifying stream,
    the eager step launching from the main stream (the form the capture uses), and the captured exchange; two streams, shared
    projections, dropout and ragged masks on, 2 MB buckets so that many collectives start in the middle of backward.
    Each mode runs in its own process: torch's NCCL watchdog thread sporadically (1 run in 8 on ROCm 7.2 / torch 2.10) polls a
    work whose end event was recorded while the step was being captured -- `hipErrorCapturedEvent`, which that thread turns into
    std::terminate (gpurun_out/cap_8.log of round 4).  That abort is an upstream race of the CAPTURED mode, not an
    ordering result: the test reports it as an expected failure of that one run instead of taking the whole suite down."""
    r = _run_rccl_script(_EXCHANGE_ORDER_SCRIPT, mode)
    assert r.returncode == 0 and "EXCHANGE-ORDER-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_two_models_interleaved_in_one_process(H):
    """the host side keeps step state in module-level variables (step id, shadow bookkeeping, deferred reduces and
    weight-gradient queues, half-reports of shared projections, the packed-sequence plan cache).  Two models of different shapes
    whose forwards and backwards interleave -- A fwd, B fwd, A bwd, B bwd; both losses in one backward() call; and each behind its
    own DataParallelStep with flat gradient buffers -- must produce exactly the gradients they produce alone."""
    from hri_emo_amd.dp import DataParallelStep
    from hri_emo_amd.train import fusion_step_loss
    torch.manual_seed(5)
    mA = H.FusionWithEmotionDecoder(d_model=256, num_emotions=5, n_heads=8, dropout=0.0).cuda().train()
    mB = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=4, num_layers_fusion=1, num_layers_decoder=3, dropout=0.0).cuda().train()
    bA = tuple(cu(t) for t in _rand_batch(3, 70, 30, 256, 61)); yA = (torch.rand(3, 5, generator=torch.Generator().manual_seed(1)) < 0.3).float().cuda()
    bB = tuple(cu(t) for t in _rand_batch(5, 40, 40, 128, 62)); yB = (torch.rand(5, 4, generator=torch.Generator().manual_seed(2)) < 0.3).float().cuda()

    def loss_of(m, b, y):
        logits, beta, _ = m(*b)
        return fusion_step_loss(logits, beta, y)

    def grads(m):
        return {n: p.grad.detach().clone() for n, p in m.named_parameters()}

    ref = {}
    for tag, m, b, y in (("A", mA, bA, yA), ("B", mB, bB, yB)):          # each model alone
        m.zero_grad(set_to_none=True)
        loss_of(m, b, y).backward()
        torch.cuda.synchronize()
        ref[tag] = grads(m)
    for order in ("separate", "joint"):
        mA.zero_grad(set_to_none=True); mB.zero_grad(set_to_none=True)
        lA = loss_of(mA, bA, yA)
        lB = loss_of(mB, bB, yB)                                         # B's forward before A's backward
        if order == "separate":
            lA.backward(); lB.backward()
        else:
            (lA + lB).backward()
        torch.cuda.synchronize()
        for tag, m in (("A", mA), ("B", mB)):
            for n, g in grads(m).items():
                assert torch.equal(g, ref[tag][n]), (order, tag, n)
    # the same through two DataParallelStep objects (flat gradient buffers, in-place accumulation, launch-boundary reduce)
    dA, dB = DataParallelStep(mA, fusion_step_loss, overlap=False), DataParallelStep(mB, fusion_step_loss, overlap=False)
    dA.step(*bA, yA); dB.step(*bB, yB)
    torch.cuda.synchronize()
    fA, fB = dA.buckets.flat.clone(), dB.buckets.flat.clone()
    for n, p in mA.named_parameters():
        assert (p.grad - ref["A"][n]).abs().max() <= 1e-6 * max(1.0, ref["A"][n].abs().max().item()), n
    dB.buckets.zero_grad(); dA.buckets.zero_grad()
    lA = loss_of(mA, bA, yA); lB = loss_of(mB, bB, yB)
    lB.backward(); lA.backward()
    torch.cuda.synchronize()
    assert torch.equal(dA.buckets.flat, fA) and torch.equal(dB.buckets.flat, fB)


def test_single_label_train_step_vs_oracle(H):
    """The IEMOCAP trainer's single_label branch (train_fusion_seq_level_decoder.py:312-314,325-326,413-414): CrossEntropyLoss
    on the logits + the beta regulariser, through hri_emo_amd.train.fusion_step_loss_single_label (one kernel) -- loss, logits and
    every parameter gradient of the whole model against the fp32 oracle with torch's own F.cross_entropy, dropout 0."""
    import torch.nn.functional as F
    from hri_emo_amd.train import fusion_step_loss_single_label
    torch.manual_seed(1234)
    kw = dict(d_model=256, num_emotions=4, n_heads=8, dropout=0.0)
    ref = O.FusionWithEmotionDecoder(**kw).train()
    m = H.FusionWithEmotionDecoder(**kw)
    m.load_state_dict(ref.state_dict())
    m.cuda().train()
    h_a, h_t, m_a, m_t = _rand_batch(5, 90, 36, 256, 13)
    labels = torch.randint(0, 4, (5,), generator=torch.Generator().manual_seed(14))

    def step(model, dev, autocast=False):
        to = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t)
        model.zero_grad()
        if autocast:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                logits, beta, _ = model(to(h_a), to(h_t), to(m_a), to(m_t))
            logits, beta = logits.float(), beta.float()
        else:
            logits, beta, _ = model(to(h_a), to(h_t), to(m_a), to(m_t))
        if dev == "cuda":
            loss = fusion_step_loss_single_label(logits, beta, to(labels))
        else:
            loss = F.cross_entropy(logits, labels) - 0.01 * (beta * (1.0 - beta)).mean()
        loss.backward()
        return loss.detach(), logits.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}

    loss_r, logits_r, gr = step(ref, "cpu")
    _, _, gy = step(ref, "cpu", autocast=True)
    loss_m, logits_m, gm = step(m, "cuda")
    close(loss_m.reshape(1), loss_r.reshape(1), what="loss"); close(logits_m, logits_r, what="logits")
    assert_per_parameter_grads(gm, gy, gr, exceptions={"beta_gate.mlp.0.weight": 0.15, "beta_gate.mlp.0.bias": 0.15}, what="single_label")


@pytest.mark.parametrize("varlen", [False, True])
def test_fused_linear_layernorm_sites_give_the_same_training_step(H, monkeypatch, varlen):
    """HRIEMO_FUSE_LN (csrc/gemm_ln.hip, opt-in: measured slower): every encoder sub-layer's output projection +
    bias + dropout + residual + LayerNorm from ONE kernel.  Same weights, same batch, same dropout seeds, fused vs separate
    launches: the kernels agree to the rounding of the row statistics, so the whole train-mode step (dropout 0.1, ragged masks;
    padded and packed rows) must agree inside the bf16 path's own error -- outputs within TOL, every parameter's gradient within twice GRAD_FLOOR (measured: 3.7 % worst, decoder FFN)."""
    from hri_emo_amd import _ops
    torch.manual_seed(1234)
    m = H.FusionWithEmotionDecoder(d_model=768, num_emotions=6, n_heads=8, dropout=0.1).cuda().train()
    B, Ta, Tt = 18, 100, 60                      # 1800 / 1080 rows: both branches above the fused path's 1024-row threshold
    h_a, h_t, m_a, m_t = _rand_batch(B, Ta, Tt, 768, 21)
    y = (torch.rand(B, 6, generator=torch.Generator().manual_seed(3)) < 0.3).float()
    args = (cu(h_a), cu(h_t), cu(m_a), cu(m_t), cu(y))
    H.set_varlen(varlen)
    try:
        res = {}
        for fused in (False, True):
            monkeypatch.setattr(_ops, "FUSE_LN", fused)
            monkeypatch.setattr(_ops, "FUSE_LN_MIN_ROWS", 512)   # packed text branch: ~810 rows; the decoder (108 rows) stays out
            calls = []
            real = _ops.proj_add_ln_fwd
            monkeypatch.setattr(_ops, "proj_add_ln_fwd", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
            torch.manual_seed(77)                # the dropout seeds come from torch's CPU generator: the same for both runs
            res[fused] = _train_step(m, *args) + (len(calls),)
            monkeypatch.setattr(_ops, "proj_add_ln_fwd", real)
    finally:
        H.set_varlen(False)
    (l0, lg0, ga0, gt0, g0, n0), (l1, lg1, ga1, gt1, g1, n1) = res[False], res[True]
    assert n0 == 0 and n1 == 2 * 6                # six post-LN sites per fusion layer took the fused kernel, the decoder none
    if not torch.equal(lg0, lg1):                 # (identical seeds are required for a comparison: checked through the outputs)
        assert float((lg0 - lg1).abs().max()) <= TOL * max(1.0, float(lg0.abs().max())), "logits"
    assert abs(float(l0) - float(l1)) <= TOL * max(1.0, abs(float(l0)))
    # two bf16 paths that differ in the rounding of a few LayerNorm outputs: every parameter within the floor any bf16 backward is
    # allowed against the oracle (GRAD_FLOOR); the gate's first Linear -- a cancellation-heavy sum over B pooled rows that moves by
    # 2-11 % between ANY two bf16 evaluations (see the dropout-exact test above) -- within its stated 15 %
    rows = sorted(((_rel(g1[n], g0[n]), n) for n in g0), reverse=True)
    gate = ("beta_gate.mlp.0.weight", "beta_gate.mlp.0.bias")
    for e, n in rows:                                    # (two evaluations, each within the floor of the oracle: twice the floor apart)
        assert e <= (0.15 if n in gate else 2 * GRAD_FLOOR), ("worst five:", rows[:5])
    assert _rel(ga1, ga0) <= GRAD_FLOOR and _rel(gt1, gt0) <= GRAD_FLOOR
    print(f"fused vs separate (varlen {varlen}): worst five {[(round(e, 4), n) for e, n in rows[:5]]}")


@pytest.mark.gpu
def test_queued_small_gradients_equal_the_unqueued_ones(H, monkeypatch):
    """hri_emo_amd._ops._DeferredWgrad: the decoder's and the gate's weight gradients AND (round 4) their bias-gradient column sums
    leave the serial chain and are issued at the end of the text branch's backward.  The column sums run the same kernels on the
    same partial sums and end in the same launch-boundary reduce: every vector gradient must hold the same bits with the queue on
    and off.  The queued weight gradients leave through the grouped GEMM launch (another tile, another summation order): equal
    to fp32 rounding.  Dropout is active: the decoder FFN's mid dropout sits between the two queued column sums."""
    from hri_emo_amd import _ops
    from hri_emo_amd.dp import DataParallelStep
    from hri_emo_amd.train import fusion_step_loss
    kw = dict(d_model=256, num_emotions=5, n_heads=8, dropout=0.1)
    b = tuple(cu(t) for t in _rand_batch(4, 90, 40, 256, 77))
    y = (torch.rand(4, 5, generator=torch.Generator().manual_seed(3)) < 0.3).float().cuda()
    torch.manual_seed(11)
    m = H.FusionWithEmotionDecoder(**kw).cuda().train()       # ONE model: dropout sites are numbered per module instance
    dp = DataParallelStep(m, fusion_step_loss, overlap=False)
    grads = {}
    for on in (True, False, True):
        monkeypatch.setattr(_ops, "DEFER_SMALL_DW", on)
        torch.manual_seed(12)                                   # the same per-call dropout seeds ...
        _ops.seed_word(torch.device("cuda", 0)).zero_()         # ... on top of the same device seed word
        dp.step(*b, y)
        torch.cuda.synchronize()
        g = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
        assert all(torch.isfinite(t).all() for t in g.values())
        if on in grads:                                         # the step itself is reproducible from its seeds
            assert all(torch.equal(g[n], grads[on][n]) for n in g)
        grads[on] = g
    assert grads[True]["beta_gate.mlp.0.bias"].abs().max() > 0 and grads[True]["emotion_decoder.layers.0.linear1.bias"].abs().max() > 0
    for n, g in grads[True].items():
        r = grads[False][n]
        if g.dim() < 2:
            assert torch.equal(g, r), (n, (g - r).abs().max().item())
        else:
            assert (g - r).abs().max() <= 2e-5 * max(1e-3, r.abs().max().item()), (n, (g - r).abs().max().item(), r.abs().max().item())
