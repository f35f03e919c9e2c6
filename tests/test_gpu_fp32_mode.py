"""GPU suite of the fp32-tolerance inference mode (HRIEMO_PRECISION=fp32 / hri_emo_amd.set_precision("fp32"); hri-emo_amd/_fp32.py,
csrc/fp32mode.hip): the reference's fp32 modules to 1e-3 on the same golden vectors the bf16 path is held to 5e-3 / 1e-2 on.

Tolerance, written here once: the mode's specification is |got - ref| <= 1e-3 * max(1, max|ref|); the tests hold every output
of the golden fixtures and of the seeded oracle comparisons (logits, beta, z, every attention map) to 1e-4 -- measured worst
case 2e-5 (profiles/r02_fp32_mode.log) -- and check the kernels themselves against float64 at 2e-5 and below."""
import math

import pytest
import torch

from conftest import load_golden
from oracle import hri_emo_oracle as O          # the checker (tests only)

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture()
def H():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import hri_emo_amd
    hri_emo_amd.set_precision("fp32")
    yield hri_emo_amd
    hri_emo_amd.set_precision("bf16")


def close(got, ref, tol=TOL, what=""):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), (what, err, ref.abs().max().item())
    return err


def cu(t):
    return None if t is None else t.cuda()


def fusion(H, d, ne, p=0.1):
    return O.closed_form_init_(H.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=p)).cuda()


# ----------------------------------------------------------------------------- kernels against float64
def test_split3_reconstructs_to_2_pow_minus_16(H):
    from hri_emo_amd import _fp32
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(300, 136, generator=g) * torch.logspace(-3, 3, 136)[None, :]).cuda()
    for layout in (0, 1):
        y = _fp32.split3(x, layout=layout).float()
        K = x.shape[1]
        hi = y[:, :K]
        mid = y[:, K:2 * K] if layout == 0 else y[:, 2 * K:]
        again = y[:, 2 * K:] if layout == 0 else y[:, K:2 * K]
        assert torch.equal(hi, again)
        assert torch.equal(hi, x.bfloat16().float())
        assert float(((hi + mid) - x).abs().max() / x.abs().max()) <= 2.0 ** -16
        rel = ((hi + mid) - x).abs() / x.abs().clamp_min(1e-30)
        assert float(rel.max()) <= 2.0 ** -15
    yr = _fp32.split3(-x.abs(), relu=True).float()
    assert float(yr.abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(200, 136, 96), (1024, 768, 768), (64, 256, 3072), (130, 2304, 768)])
def test_linear_x3_against_float64(H, M, N, K):
    from hri_emo_amd import _fp32, _ops
    g = torch.Generator().manual_seed(M + N)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    sh = _ops.Shadows()
    wp, bp = torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda())
    y = _fp32.linear(x.cuda(), sh, wp, bp)
    ref = x.double() @ w.double().t() + b.double()
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err <= 2e-5, err
    # what the bf16 operands alone would give on the same problem, for scale (printed with -s)
    e16 = float(((x.bfloat16().double() @ w.bfloat16().double().t() + b.double()) - ref).abs().max() / ref.abs().max())
    print(f"x3 linear {M}x{N}x{K}: max err / max|ref| = {err:.2e} (bf16 operands: {e16:.2e})")
    y2 = _fp32.linear(x.cuda(), sh, wp, bp, rows=(8, 72), relu_in=True)
    ref2 = x.double().clamp(min=0) @ w.double()[8:72].t() + b.double()[8:72]
    assert float((y2.double().cpu() - ref2).abs().max() / ref2.abs().max()) <= 2e-5


@pytest.mark.parametrize("B,H_,Lq,Lk,hd,masked", [(2, 8, 100, 40, 96, True), (3, 4, 33, 130, 64, True), (2, 2, 400, 128, 128, False),
                                                   (1, 8, 6, 77, 16, True), (2, 4, 64, 64, 32, False)])
def test_attention_f32_against_float64(H, B, H_, Lq, Lk, hd, masked):
    from hri_emo_amd import _fp32
    g = torch.Generator().manual_seed(Lq + Lk)
    d = H_ * hd
    q, kv = torch.randn(B * Lq, d, generator=g) * 2.0, torch.randn(B * Lk, 2 * d, generator=g)
    kpm = None
    if masked:
        lens = torch.randint(1, Lk + 1, (B,), generator=g)
        kpm = torch.arange(Lk)[None, :] >= lens[:, None]
    qd, kvd = q.cuda(), kv.cuda()
    kpm_d = kpm.cuda().view(torch.uint8) if kpm is not None else None
    o, lse = _fp32.attn(qd, kvd[:, :d], kvd[:, d:], B, H_, Lq, Lk, hd, kpm_d, want_lse=True)
    p = _fp32.probs(qd, kvd[:, :d], B, H_, Lq, Lk, hd, kpm_d, lse)
    q4 = q.double().view(B, Lq, H_, hd).transpose(1, 2)
    k4 = kv[:, :d].double().reshape(B, Lk, H_, hd).transpose(1, 2)
    v4 = kv[:, d:].double().reshape(B, Lk, H_, hd).transpose(1, 2)
    s = q4 @ k4.transpose(-1, -2) / math.sqrt(hd)
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
    pr = torch.softmax(s, -1)
    ref = (pr @ v4).transpose(1, 2).reshape(B * Lq, d)
    assert float((o.double().cpu() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert float((p.double().cpu() - pr.mean(1)).abs().max()) <= 2e-6
    assert float((lse.double().cpu() - torch.logsumexp(s, -1)).abs().max()) <= 2e-5
    if kpm is not None:
        assert float(p.cpu()[kpm[:, None, :].expand(B, Lq, Lk)].abs().max()) == 0.0       # PAD key columns exactly 0


def test_add_ln_f32_and_gate_kernels_against_float64(H):
    from hri_emo_amd import _fp32, _ops, _lib
    g = torch.Generator().manual_seed(3)
    for (M, d) in [(100, 768), (7, 128), (33, 2048)]:
        x, r = torch.randn(M, d, generator=g) * 3, torch.randn(M, d, generator=g)
        gm, bt = torch.randn(d, generator=g), torch.randn(d, generator=g)
        y16, y32 = _fp32.add_ln(x.cuda(), r.cuda(), gm.cuda(), bt.cuda())
        ref = torch.nn.functional.layer_norm((x + r).double(), (d,), gm.double(), bt.double(), 1e-5)
        assert float((y32.double().cpu() - ref).abs().max()) <= 5e-6 * max(1.0, float(ref.abs().max()))
        assert torch.equal(y16.float().cpu(), y32.cpu().bfloat16().float())
        _, y = _fp32.add_ln(x.cuda(), None, gm.cuda(), bt.cuda(), want16=False)
        ref = torch.nn.functional.layer_norm(x.double(), (d,), gm.double(), bt.double(), 1e-5)
        assert float((y.double().cpu() - ref).abs().max()) <= 5e-6 * max(1.0, float(ref.abs().max()))
    B, La, Lt, d = 3, 50, 20, 128
    A, T = torch.randn(B, La, d, generator=g), torch.randn(B, Lt, d, generator=g)
    ma = torch.arange(La)[None] >= torch.tensor([50, 1, 30])[:, None]
    pooled = torch.empty(B, d, device="cuda")
    Ad, Td, mad = A.cuda(), T.cuda(), ma.cuda().view(torch.uint8)      # device operands stay referenced across the launches
    _lib.call("hriemo_masked_mean_f32", _ops._p(Ad), _ops._p(mad), _ops._p(pooled), B, La, d, _ops._stream())
    keep = (~ma).double()[:, :, None]
    ref = (A.double() * keep).sum(1) / keep.sum(1).clamp(min=1.0)
    assert float((pooled.double().cpu() - ref).abs().max()) <= 1e-5
    w = torch.rand(B, d, generator=g)
    H32 = torch.empty(B, Lt, d, device="cuda"); H16 = torch.empty(B, Lt, d, device="cuda", dtype=torch.bfloat16)
    wd = w.cuda()
    _lib.call("hriemo_fuse_f32", _ops._p(wd), _ops._p(Ad), La, _ops._p(Td), Lt, _ops._p(H32), _ops._p(H16), B, Lt, d, _ops._stream())
    ref = w[:, None, :] * A[:, :Lt] + (1 - w[:, None, :]) * T
    assert float((H32.cpu() - ref).abs().max()) <= 1e-6
    assert torch.equal(H16.float().cpu(), H32.cpu().bfloat16().float())


# ----------------------------------------------------------------------------- modules against the golden vectors, 1e-3
@pytest.mark.parametrize("name", ["cfg1_eval_nomask", "cfg1_eval_ragged", "cfg1_eval_2d_inputs"])
def test_fusion_eval_vs_golden_fp32(H, name):
    g = load_golden(name)
    m = fusion(H, 128, 4).eval()
    with torch.no_grad():
        logits, beta, z = m(cu(g["h_a"]), cu(g["h_t"]), cu(g.get("mask_a")), cu(g.get("mask_t")))
    assert logits.dtype == torch.float32 and z.dtype == torch.float32
    errs = [close(logits, g["logits"], what="logits"), close(beta, g["beta"], what="beta"), close(z, g["z"], what="z")]
    print(f"{name}: max abs err logits {errs[0]:.2e} beta {errs[1]:.2e} z {errs[2]:.2e}")


@pytest.mark.parametrize("name,d,ne", [("cfg1_eval_ragged", 128, 4), ("hd96_eval_ragged", 768, 6)])
def test_fusion_attention_maps_vs_golden_fp32(H, name, d, ne):
    g = load_golden(name)
    m = fusion(H, d, ne).eval()
    with torch.no_grad():
        logits, beta, z, pack = m(cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]), return_attention=True)
    close(logits, g["logits"], what="logits"); close(z, g["z"], what="z"); close(beta, g["beta"], what="beta")
    assert len(pack["encoder"]) == 2 and len(pack["decoder"]) == 2
    worst = 0.0
    for li, maps in enumerate(pack["encoder"]):
        for k, v in maps.items():
            worst = max(worst, close(v, g[f"enc.{li}.{k}"], what=f"enc.{li}.{k}"))
    for li, v in enumerate(pack["decoder"]):
        worst = max(worst, close(v, g[f"dec.{li}"], what=f"dec.{li}"))
    print(f"{name}: worst attention-map error {worst:.2e} (the bf16 path is held to 2e-2 on the same fixture)")
    w = pack["encoder"][-1]["audio_queries_text"].cpu()
    assert (w[g["mask_t"][:, None, :].expand_as(w)] == 0).all()
    close(w.sum(-1), torch.ones(w.shape[:-1]), 1e-5, "rows sum to one")


def test_cfg2_shape_vs_reference_golden_fp32(H):
    from conftest import cfg2_seeded_inputs
    g = load_golden("cfg2_seeded")
    h_a, h_t, m_a, m_t = cfg2_seeded_inputs(g)
    m = fusion(H, 768, 6, p=0.0).eval()
    with torch.no_grad():
        logits, beta, z = m(cu(h_a), cu(h_t), cu(m_a), cu(m_t))
    e = [close(logits, g["logits"], what="logits"), close(beta, g["beta"], what="beta"), close(z, g["z"], what="z")]
    print(f"cfg-2 shape vs the reference golden, fp32 mode: logits {e[0]:.2e} beta {e[1]:.2e} z {e[2]:.2e}")


def test_fusion_allpad_row_nan_only_for_that_sample_fp32(H):
    g = load_golden("cfg1_eval_allpad_row")
    m = fusion(H, 128, 4).eval()
    with torch.no_grad():
        logits, beta, z = m(cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]))
    logits = logits.cpu()
    assert torch.equal(torch.isnan(logits), torch.isnan(g["logits"]))
    ok = ~torch.isnan(g["logits"])
    close(logits[ok], g["logits"][ok])


def test_components_vs_golden_fp32(H):
    g = load_golden("block_eval_ragged")
    blk = O.closed_form_init_(H.CrossModalBlock(128, 8, 0.1)).cuda().eval()
    with torch.no_grad():
        oa, ot, maps = blk(cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]), return_attention=True)
    assert oa.dtype == torch.float32
    close(oa, g["out_a"], what="out_a"); close(ot, g["out_t"], what="out_t")
    for k, v in maps.items():
        close(v, g["map." + k], what=k)
    gg = load_golden("gate_eval_ragged")
    gate = O.closed_form_init_(H.BetaGate(128, 32)).cuda().eval()
    with torch.no_grad():
        hf, beta = gate(cu(gg["h_a"]), cu(gg["h_t"]), cu(gg["mask_a"]), cu(gg["mask_t"]))
    close(hf, gg["h_fusion"], what="h_fusion"); close(beta, gg["beta"], what="beta")
    gg = load_golden("gate_eval_equal_len_nomask")
    with torch.no_grad():
        hf, beta = gate(cu(gg["h_a"]), cu(gg["h_t"]))
    close(hf, gg["h_fusion"], what="h_fusion eq"); close(beta, gg["beta"], what="beta eq")
    gd = load_golden("decoder_eval_ragged")
    dec = O.closed_form_init_(H.EmotionDecoder(128, 5, 8, 2, 64, 0.1)).cuda().eval()
    with torch.no_grad():
        z, logits, maps = dec(cu(gd["memory"]), cu(gd["mask"]), return_attention=True)
    close(z, gd["z"], what="z"); close(logits, gd["logits"], what="logits")
    for i, v in enumerate(maps):
        close(v, gd[f"map.{i}"], what=f"dec map {i}")


def test_mosei_wrapper_vs_golden_fp32(H):
    g = load_golden("mosei_eval_train")
    ref = O.closed_form_init_(O.MoseiFusionWithEmotionDecoder(d_audio=74, d_text=300))
    m = H.MoseiFusionWithEmotionDecoder(d_audio=74, d_text=300)
    m.load_state_dict(ref.state_dict(), strict=True)
    m.cuda().eval()
    with torch.no_grad():
        logits, beta, z = m(cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]))
    close(logits, g["logits"], what="logits"); close(beta, g["beta"], what="beta"); close(z, g["z"], what="z")


def _rand_batch(B, Ta, Tt, d, seed):
    g = torch.Generator().manual_seed(seed)
    h_a, h_t = torch.randn(B, Ta, d, generator=g), torch.randn(B, Tt, d, generator=g)
    la = torch.randint(max(1, Ta // 2), Ta + 1, (B,), generator=g)
    lt = torch.randint(max(1, Tt // 2), Tt + 1, (B,), generator=g)
    return h_a, h_t, torch.arange(Ta)[None] >= la[:, None], torch.arange(Tt)[None] >= lt[:, None]


@pytest.mark.parametrize("B,Ta,Tt,d,ne,lf,ld", [
    (2, 400, 128, 768, 6, 2, 2),          # BASELINE configs[1]/[2]: the headline shape
    (2, 1000, 50, 768, 6, 2, 2),          # BASELINE configs[3]: MOSEI shape
    (2, 400, 128, 1024, 7, 4, 2),         # BASELINE configs[4] dimensions
    (1, 1, 1, 128, 1, 1, 1), (3, 17, 5, 128, 3, 2, 2), (2, 64, 64, 512, 5, 1, 1),
])
def test_fusion_vs_oracle_seeded_fp32(H, B, Ta, Tt, d, ne, lf, ld):
    torch.manual_seed(1234)
    kw = dict(d_model=d, num_emotions=ne, n_heads=8, num_layers_fusion=lf, num_layers_decoder=ld, dropout=0.1)
    ref = O.FusionWithEmotionDecoder(**kw).eval()
    m = H.FusionWithEmotionDecoder(**kw)
    m.load_state_dict(ref.state_dict())
    m.cuda().eval()
    h_a, h_t, m_a, m_t = _rand_batch(B, Ta, Tt, d, 17)
    with torch.no_grad():
        lr, br, zr, pr = ref(h_a, h_t, m_a, m_t, return_attention=True)
        lg, bg, zg, pg = m(cu(h_a), cu(h_t), cu(m_a), cu(m_t), return_attention=True)
    e = [close(lg, lr, what="logits"), close(bg, br, what="beta"), close(zg, zr, what="z")]
    worst = 0.0
    for mg, mr in zip(pg["encoder"], pr["encoder"]):
        for k in mr:
            worst = max(worst, close(mg[k], mr[k], what=k))
    for vg, vr in zip(pg["decoder"], pr["decoder"]):
        worst = max(worst, close(vg, vr, what="dec map"))
    print(f"fp32 mode vs oracle B{B} Ta{Ta} Tt{Tt} d{d}: logits {e[0]:.2e} beta {e[1]:.2e} z {e[2]:.2e} maps {worst:.2e}")


def test_fp32_mode_is_forward_only_and_says_so(H):
    m = fusion(H, 128, 4).eval()
    g = load_golden("cfg1_eval_nomask")
    with pytest.raises(RuntimeError, match="inference mode"):
        m(cu(g["h_a"]), cu(g["h_t"]))                       # parameters require grad, autograd is recording
    with pytest.raises(ValueError):
        H.set_precision("fp64")
    # bf16 inputs are accepted (exact upcast) and come back as bf16
    with torch.no_grad():
        logits, beta, z = m(cu(g["h_a"]).bfloat16(), cu(g["h_t"]).bfloat16())
    assert z.dtype == torch.bfloat16 and logits.dtype == torch.float32
