"""GPU suite, kernel level, part 2: the beta-gate / pooling / fusion / dropout row kernels, each called THROUGH THE C ABI
(include/hriemo.h, ctypes) and compared with plain torch fp32 math on the same inputs.

Reference arithmetic: models/beta_gate_tacfn.py:6-24 (masked_mean), :79-84 (LayerNorm + pools), :87-95 (gate input, sigmoid,
beta), :98-116 (fusion over the first L positions); models/beta_gate.py:6-32,97-112 (legacy scalar gate);
models/emotion_decoder.py:58 (mid-FFN dropout).  Tolerances are stated per check: fp32 results 1e-5-ish, bf16 outputs one
bf16 ulp of the value range (2^-8 relative)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import hashrng

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import hri_emo_amd  # noqa: F401
    from hri_emo_amd import _lib
    return _lib


def P(t):
    return None if t is None else t.data_ptr()


def ST():
    return torch.cuda.current_stream().cuda_stream


def f32(*shape):
    return torch.empty(shape, dtype=torch.float32, device="cuda")


def bf(*shape):
    return torch.empty(shape, dtype=torch.bfloat16, device="cuda")


def relerr(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def ragged_mask(B, L, g, lo_frac=0.4):
    lens = torch.randint(max(1, int(L * lo_frac)), L + 1, (B,), generator=g)
    return torch.arange(L)[None, :] >= lens[:, None]


GATE_CASES = [  # B, La, Lt, d, masked, twin
    (3, 40, 17, 128, True, False),
    (2, 400, 128, 768, True, True),
    (4, 33, 33, 256, False, True),        # equal lengths, no masks (mask pointers NULL)
    (2, 70, 5, 1024, True, True),
    (1, 1, 1, 128, False, False),
]


@pytest.mark.parametrize("B,La,Lt,d,masked,twin", GATE_CASES)
def test_gate_forward_chain_through_c_abi(lib, B, La, Lt, d, masked, twin):
    """hriemo_ln_pool_fwd (x2) -> hriemo_gate_input -> hriemo_sigmoid_beta -> hriemo_fuse_fwd against
    LayerNorm / masked mean / cat[a,t,|a-t|,a*t] / sigmoid / w*A+(1-w)*T in fp32 (beta_gate_tacfn.py:79-116)."""
    g = torch.Generator().manual_seed(1000 + La + d)
    L = Lt
    xa32, xt32 = torch.randn(B, La, d, generator=g) * 1.3 + 0.2, torch.randn(B, Lt, d, generator=g)
    xa, xt = xa32.bfloat16(), xt32.bfloat16()
    ga, ba = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    gt, bt = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    ma = ragged_mask(B, La, g) if masked else None
    mt = ragged_mask(B, Lt, g) if masked else None
    src_a = xa32 if twin else xa.float()            # the fp32 twin wins over the bf16 copy when given
    src_t = xt32 if twin else xt.float()
    An_r, Tn_r = F.layer_norm(src_a, (d,), ga, ba, 1e-5), F.layer_norm(src_t, (d,), gt, bt, 1e-5)

    def mmean(x, m):
        if m is None:
            return x.mean(1), torch.full((x.shape[0],), float(x.shape[1]))
        v = (~m).float()
        return (x * v[..., None]).sum(1) / v.sum(1).clamp(min=1)[:, None], v.sum(1).clamp(min=1)

    ap_r, ca_r = mmean(An_r, ma)
    tp_r, ct_r = mmean(Tn_r, mt)

    nca, nct = lib.lib().hriemo_pool_chunks(La), lib.lib().hriemo_pool_chunks(Lt)
    An, Tn = bf(B, L, d), bf(B, L, d)
    mean_a, rstd_a, mean_t, rstd_t = f32(B * La), f32(B * La), f32(B * Lt), f32(B * Lt)
    pa, pt = f32(B, nca, d), f32(B, nct, d)
    xa_d, xt_d = xa.cuda(), xt.cuda()
    xa32_d, xt32_d = (xa32.cuda(), xt32.cuda()) if twin else (None, None)
    ma_d = ma.cuda().view(torch.uint8) if ma is not None else None
    mt_d = mt.cuda().view(torch.uint8) if mt is not None else None
    ga_d, ba_d, gt_d, bt_d = ga.cuda(), ba.cuda(), gt.cuda(), bt.cuda()
    lib.call("hriemo_ln_pool_fwd", P(xa_d), P(xa32_d), P(ma_d), P(ga_d), P(ba_d), P(An), P(mean_a), P(rstd_a), P(pa), B, La, L, d,
             1e-5, ST())
    lib.call("hriemo_ln_pool_fwd", P(xt_d), P(xt32_d), P(mt_d), P(gt_d), P(bt_d), P(Tn), P(mean_t), P(rstd_t), P(pt), B, Lt, L, d,
             1e-5, ST())
    assert relerr(An, An_r[:, :L]) <= 2 ** -8 and relerr(Tn, Tn_r[:, :L]) <= 2 ** -8             # bf16 outputs
    assert (mean_a.cpu() - src_a.mean(-1).reshape(-1)).abs().max() <= 1e-5
    var = src_a.var(-1, unbiased=False).reshape(-1)
    assert relerr(rstd_a, (var + 1e-5).rsqrt()) <= 1e-5

    gin, a_pool, t_pool, cnt = bf(B, 4 * d), f32(B, d), f32(B, d), f32(B, 2)
    lib.call("hriemo_gate_input", P(pa), P(pt), P(ma_d), P(mt_d), B, La, Lt, d, P(gin), P(a_pool), P(t_pool), P(cnt), ST())
    assert (a_pool.cpu() - ap_r).abs().max() <= 2e-5 * max(1.0, ap_r.abs().max().item())      # fp32 pools of fp32 LN values
    assert (t_pool.cpu() - tp_r).abs().max() <= 2e-5 * max(1.0, tp_r.abs().max().item())
    assert torch.equal(cnt.cpu(), torch.stack([ca_r, ct_r], 1))
    gin_r = torch.cat([ap_r, tp_r, (ap_r - tp_r).abs(), ap_r * tp_r], -1)
    assert (gin.float().cpu() - gin_r).abs().max() <= 2 ** -8 * max(1.0, gin_r.abs().max().item())

    pre = (torch.randn(B, d, generator=g) * 2).cuda()
    w, beta = f32(B, d), f32(B, 1)
    lib.call("hriemo_sigmoid_beta", P(pre), P(w), P(beta), B, d, ST())
    assert (w.cpu() - torch.sigmoid(pre.cpu())).abs().max() <= 2e-6
    assert (beta.cpu() - torch.sigmoid(pre.cpu()).mean(-1, keepdim=True)).abs().max() <= 2e-6

    H = bf(B, L, d)
    lib.call("hriemo_fuse_fwd", P(w), P(An), P(Tn), P(H), B, L, d, ST())
    H_r = w.cpu()[:, None, :] * An.float().cpu() + (1 - w.cpu()[:, None, :]) * Tn.float().cpu()   # from the kernel's own bf16 A, T
    assert (H.float().cpu() - H_r).abs().max() <= 2 ** -8 * max(1.0, H_r.abs().max().item())


@pytest.mark.parametrize("B,La,Lt,d,masked,twin", GATE_CASES)
def test_gate_backward_chain_through_c_abi(lib, B, La, Lt, d, masked, twin):
    """hriemo_fuse_bwd_dw -> hriemo_gate_dpre -> hriemo_gate_input_bwd -> hriemo_ln_pool_bwd (both modalities) against
    torch autograd of the same fp32 graph, fed with the same upstream gradients (dH, dbeta, d gate_in)."""
    g = torch.Generator().manual_seed(2000 + La + d)
    L = Lt
    xa32, xt32 = torch.randn(B, La, d, generator=g) * 1.3 + 0.2, torch.randn(B, Lt, d, generator=g)
    xa, xt = xa32.bfloat16(), xt32.bfloat16()
    ga, gt = 1 + 0.1 * torch.randn(d, generator=g), 1 + 0.1 * torch.randn(d, generator=g)
    ba, bt = torch.zeros(d), torch.zeros(d)
    ma = ragged_mask(B, La, g) if masked else None
    mt = ragged_mask(B, Lt, g) if masked else None
    dH = torch.randn(B, L, d, generator=g).bfloat16()
    dbeta = torch.randn(B, 1, generator=g)
    dgin = (torch.randn(B, 4 * d, generator=g) * 0.5).bfloat16()
    wv = torch.sigmoid(torch.randn(B, d, generator=g))

    # ---- torch reference graph: (x_a, x_t, gamma/beta, w, gate_in) -> scalar
    src_a = (xa32 if twin else xa.float()).clone().requires_grad_(True)
    src_t = (xt32 if twin else xt.float()).clone().requires_grad_(True)
    ga_r, ba_r = ga.clone().requires_grad_(True), ba.clone().requires_grad_(True)
    gt_r, bt_r = gt.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    w_r = wv.clone().requires_grad_(True)
    An_r, Tn_r = F.layer_norm(src_a, (d,), ga_r, ba_r, 1e-5), F.layer_norm(src_t, (d,), gt_r, bt_r, 1e-5)

    def mmean(x, m):
        if m is None:
            return x.mean(1)
        v = (~m).float()
        return (x * v[..., None]).sum(1) / v.sum(1).clamp(min=1)[:, None]

    ap_r, tp_r = mmean(An_r, ma), mmean(Tn_r, mt)
    gin_r = torch.cat([ap_r, tp_r, (ap_r - tp_r).abs(), ap_r * tp_r], -1)
    H_r = w_r[:, None, :] * An_r[:, :L] + (1 - w_r[:, None, :]) * Tn_r[:, :L]
    obj = (H_r * dH.float()).sum() + (w_r.mean(-1, keepdim=True) * dbeta).sum() + (gin_r * dgin.float()).sum()
    obj.backward()

    # ---- the kernels, forward first (saved tensors), then every backward entry point
    nca, nct, ncl = (lib.lib().hriemo_pool_chunks(x) for x in (La, Lt, L))
    An, Tn = bf(B, L, d), bf(B, L, d)
    mean_a, rstd_a, mean_t, rstd_t = f32(B * La), f32(B * La), f32(B * Lt), f32(B * Lt)
    pa, pt = f32(B, nca, d), f32(B, nct, d)
    xa_d, xt_d = xa.cuda(), xt.cuda()
    xa32_d, xt32_d = (xa32.cuda(), xt32.cuda()) if twin else (None, None)
    ma_d = ma.cuda().view(torch.uint8) if ma is not None else None
    mt_d = mt.cuda().view(torch.uint8) if mt is not None else None
    ga_d, ba_d, gt_d, bt_d = ga.cuda(), ba.cuda(), gt.cuda(), bt.cuda()
    lib.call("hriemo_ln_pool_fwd", P(xa_d), P(xa32_d), P(ma_d), P(ga_d), P(ba_d), P(An), P(mean_a), P(rstd_a), P(pa), B, La, L, d,
             1e-5, ST())
    lib.call("hriemo_ln_pool_fwd", P(xt_d), P(xt32_d), P(mt_d), P(gt_d), P(bt_d), P(Tn), P(mean_t), P(rstd_t), P(pt), B, Lt, L, d,
             1e-5, ST())
    gin, a_pool, t_pool, cnt = bf(B, 4 * d), f32(B, d), f32(B, d), f32(B, 2)
    lib.call("hriemo_gate_input", P(pa), P(pt), P(ma_d), P(mt_d), B, La, Lt, d, P(gin), P(a_pool), P(t_pool), P(cnt), ST())

    dH_d, w_d, dbeta_d, dgin_d = dH.cuda(), wv.cuda(), dbeta.cuda().contiguous(), dgin.cuda()
    part = f32(B, ncl, d)
    lib.call("hriemo_fuse_bwd_dw", P(dH_d), P(An), P(Tn), P(part), B, L, d, ST())
    dw_ref = (dH.float() * (An.float().cpu() - Tn.float().cpu())).sum(1)                 # from the kernel's own bf16 A, T
    assert (part.sum(1).cpu() - dw_ref).abs().max() <= 1e-4 * max(1.0, dw_ref.abs().max().item())
    dpre = bf(B, d)
    lib.call("hriemo_gate_dpre", P(part), L, P(dbeta_d), P(w_d), P(dpre), B, d, ST())
    dpre_ref = (dw_ref + dbeta / d) * wv * (1 - wv)
    assert (dpre.float().cpu() - dpre_ref).abs().max() <= 2 ** -8 * max(1e-3, dpre_ref.abs().max().item())
    # against autograd: dw of the fp32 graph (A, T unrounded) -- bf16 A/T cost ~2^-8 relative per term
    assert relerr(part.sum(1) + dbeta_d / d, w_r.grad) <= 2e-2

    da, dt = f32(B, d), f32(B, d)
    lib.call("hriemo_gate_input_bwd", P(dgin_d), P(a_pool), P(t_pool), P(cnt), P(da), P(dt), B, d, ST())
    a_, t_, gi = a_pool.cpu(), t_pool.cpu(), dgin.float()
    sg = torch.sign(a_ - t_)
    da_ref = (gi[:, :d] + sg * gi[:, 2 * d:3 * d] + t_ * gi[:, 3 * d:]) / cnt.cpu()[:, :1]
    dt_ref = (gi[:, d:2 * d] - sg * gi[:, 2 * d:3 * d] + a_ * gi[:, 3 * d:]) / cnt.cpu()[:, 1:]
    assert (da.cpu() - da_ref).abs().max() <= 1e-5 * max(1.0, da_ref.abs().max().item())
    assert (dt.cpu() - dt_ref).abs().max() <= 1e-5 * max(1.0, dt_ref.abs().max().item())

    dxa, dxt = bf(B, La, d), bf(B, Lt, d)
    dga, dba, dgt, dbt = f32(d), f32(d), f32(d), f32(d)
    wsb = max(lib.lib().hriemo_ln_pool_bwd_workspace_bytes(B, La, d), lib.lib().hriemo_ln_pool_bwd_workspace_bytes(B, Lt, d))
    ws = f32(wsb // 4 + 16)
    lib.call("hriemo_ln_pool_bwd", P(dH_d), L, P(w_d), 1, P(da), P(ma_d), P(xa_d), P(xa32_d), P(ga_d), P(mean_a), P(rstd_a), P(dxa),
             P(dga), P(dba), 0, B, La, d, P(ws), ST())
    lib.call("hriemo_ln_pool_bwd", P(dH_d), L, P(w_d), 0, P(dt), P(mt_d), P(xt_d), P(xt32_d), P(gt_d), P(mean_t), P(rstd_t), P(dxt),
             P(dgt), P(dbt), 0, B, Lt, d, P(ws), ST())
    # dX: bf16 output of O(1) values; everything upstream of it is fp32 here
    for name, got, ref in (("dxa", dxa, src_a.grad), ("dxt", dxt, src_t.grad)):
        err = (got.float().cpu() - ref).abs().max().item()
        assert err <= 1.5 * 2 ** -8 * max(1.0, ref.abs().max().item()), (name, err, ref.abs().max().item())
    for name, got, ref in (("dgamma_a", dga, ga_r.grad), ("dbeta_a", dba, ba_r.grad), ("dgamma_t", dgt, gt_r.grad),
                           ("dbeta_t", dbt, bt_r.grad)):
        err = (got.cpu() - ref).abs().max().item()
        assert err <= 2e-4 * max(1.0, ref.abs().max().item()), (name, err, ref.abs().max().item())


@pytest.mark.parametrize("B,L,Lf,d,masked", [(3, 40, 17, 128, True), (2, 130, 130, 768, False), (4, 9, 4, 256, True)])
def test_legacy_scalar_gate_kernels_through_c_abi(lib, B, L, Lf, d, masked):
    """hriemo_masked_mean_fwd, hriemo_rowsum_f32 and hriemo_scalar_gate_dx (models/beta_gate.py:6-32,97-112)."""
    g = torch.Generator().manual_seed(3000 + L)
    X = torch.randn(B, L, d, generator=g).bfloat16()
    m = ragged_mask(B, L, g) if masked else None
    v = (~m).float() if m is not None else torch.ones(B, L)
    pooled_r = (X.float() * v[..., None]).sum(1) / v.sum(1).clamp(min=1)[:, None]
    m_d = m.cuda().view(torch.uint8) if m is not None else None
    pooled, cnt = f32(B, d), f32(B)
    X_d = X.cuda()              # device operands live in variables: a temporary inside the argument list is freed (and its
    lib.call("hriemo_masked_mean_fwd", P(X_d), P(m_d), P(pooled), P(cnt), B, L, d, ST())   # address reused) before the launch
    assert (pooled.cpu() - pooled_r).abs().max() <= 2e-6 * max(1.0, pooled_r.abs().max().item())
    assert torch.equal(cnt.cpu(), v.sum(1).clamp(min=1))

    x = torch.randn(B, 777, generator=g)
    out, x_d = f32(B), x.cuda()
    lib.call("hriemo_rowsum_f32", P(x_d), P(out), B, 777, ST())
    assert (out.cpu() - x.sum(1)).abs().max() <= 1e-4

    dH = torch.randn(B, Lf, d, generator=g).bfloat16()
    beta = torch.sigmoid(torch.randn(B, generator=g))
    dpool = torch.randn(B, d, generator=g)
    dH_d, beta_d, dpool_d = dH.cuda(), beta.cuda(), dpool.cuda()
    for is_a in (1, 0):
        dX = bf(B, L, d)
        lib.call("hriemo_scalar_gate_dx", P(dH_d), Lf, P(beta_d), is_a, P(dpool_d), P(cnt), P(m_d), P(dX), B, L, d, ST())
        coef = beta if is_a else 1 - beta
        ref = v[..., None] * dpool[:, None, :] / cnt.cpu()[:, None, None]
        ref[:, :Lf] += coef[:, None, None] * dH.float()
        assert (dX.float().cpu() - ref).abs().max() <= 2 ** -8 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,p,roff", [(37, 128, 0.1, 0), (384, 2048, 0.1, 12345), (5, 8, 0.5, 7), (64, 3072, 0.0, 0)])
def test_dropout_kernel_exact_vs_hash_replica(lib, M, N, p, roff):
    """hriemo_dropout_bf16 (the decoder's mid-FFN dropout, emotion_decoder.py:58): kept elements are x/(1-p) rounded to bf16,
    dropped ones exactly 0, with the keep mask of the host replica of the counter hash -- bit-exact; applying it twice
    (forward on h, backward on dh) uses the same mask."""
    g = torch.Generator().manual_seed(4000 + M)
    X = torch.randn(M, N, generator=g).bfloat16()
    seed, site = 2024_10_04, 9
    Y, X_d = bf(M, N), X.cuda()
    lib.call("hriemo_dropout_bf16", P(X_d), P(Y), M, N, float(p), seed, None, site, roff, ST())
    if p == 0.0:
        assert torch.equal(Y.cpu(), X)
        return
    keep = torch.from_numpy(hashrng.rows_mask(seed, site, M, N, p, roff))
    ref = torch.where(keep, (X.float() * np.float32(hashrng.inv_keep(p))).bfloat16(), torch.zeros((), dtype=torch.bfloat16))
    assert torch.equal(Y.cpu(), ref)
    assert abs(float((~keep).float().mean()) - p) < 0.05 + 2.0 / (M * N) ** 0.5
    # the device seed word is ADDED to the immediate seed (graph replays draw fresh masks): same as seed+delta
    word = torch.tensor([977], dtype=torch.int64, device="cuda")
    Y2, Y3 = bf(M, N), bf(M, N)
    lib.call("hriemo_dropout_bf16", P(X_d), P(Y2), M, N, float(p), seed, P(word), site, roff, ST())
    lib.call("hriemo_dropout_bf16", P(X_d), P(Y3), M, N, float(p), seed + 977, None, site, roff, ST())
    assert torch.equal(Y2, Y3) and not torch.equal(Y2, Y)


@pytest.mark.parametrize("B,Ne,mode,coef,pw,scale", [(64, 6, 1, 0.01, False, 1.0), (8, 4, 2, 0.05, True, 0.25), (3, 7, 0, 0.0, True, 1.0),
                                                      (512, 6, 2, 0.01, False, 0.5), (1, 1, 1, 0.01, False, 1.0)])
def test_fusion_loss_kernel_value_and_gradients(lib, B, Ne, mode, coef, pw, scale):
    """hriemo_fusion_loss through the C ABI and through the autograd Function against torch's BCEWithLogitsLoss(pos_weight) +
    the two trainers' beta regularisers (train_fusion_seq_level_decoder.py:318-326; train_mosei_...:340-347,385-387,569)."""
    from hri_emo_amd import _ops
    g = torch.Generator().manual_seed(B * 10 + Ne)
    x = (torch.randn(B, Ne, generator=g) * 3).requires_grad_(True)
    y = (torch.rand(B, Ne, generator=g) < 0.3).float()
    beta = torch.rand(B, 1, generator=g).requires_grad_(True)
    with torch.no_grad():
        if B >= 3:
            beta[0, 0] = 1e-9           # below the clamp: zero gradient there (the upper clamp 1 - 1e-8 is 1.0 in fp32: a beta of
                                        # exactly 1 is NaN in the reference too, and a mean of sigmoids never reaches it)
    posw = (0.5 + 3 * torch.rand(Ne, generator=g)) if pw else None
    ref = F.binary_cross_entropy_with_logits(x, y, pos_weight=posw)
    if mode == 1:
        ref = ref - coef * (beta * (1 - beta)).mean()
    elif mode == 2:
        b = torch.clamp(beta, 1e-8, 1 - 1e-8)
        ref = ref + coef * (-(b * torch.log(b) + (1 - b) * torch.log(1 - b))).mean()
    ref = ref * scale
    ref.backward()
    xd, bd = x.detach().cuda().requires_grad_(True), beta.detach().cuda().requires_grad_(True)
    loss = _ops.FusionLossFn.apply(xd, bd if mode else None, y.cuda(), posw.cuda() if pw else None, mode, coef, scale)
    (loss * 3.0).backward()                             # an outer factor (GradScaler) must flow through
    assert abs(float(loss) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    assert (xd.grad.cpu() / 3.0 - x.grad).abs().max() <= 2e-6 * max(1.0, x.grad.abs().max().item())
    if mode:
        assert (bd.grad.cpu() / 3.0 - beta.grad).abs().max() <= 1e-5 * max(1e-3, beta.grad.abs().max().item())


@pytest.mark.parametrize("B,C,mode,coef,scale", [(64, 4, 1, 0.01, 1.0), (8, 6, 0, 0.0, 1.0), (300, 7, 2, 0.05, 0.5), (1, 2, 1, 0.01, 1.0), (5, 1, 1, 0.01, 1.0)])
def test_fusion_loss_ce_kernel_value_and_gradients(lib, B, C, mode, coef, scale):
    """hriemo_fusion_loss_ce (single-label branch of the IEMOCAP trainer, train_fusion_seq_level_decoder.py:312-314,325-326 with
    criterion nn.CrossEntropyLoss() :413-414) against torch.nn.functional.cross_entropy + the beta regulariser, value and both
    gradients, through the autograd Function; and through hri_emo_amd.train.fusion_step_loss_single_label."""
    from hri_emo_amd import _ops
    from hri_emo_amd.train import fusion_step_loss_single_label
    g = torch.Generator().manual_seed(B * 10 + C)
    x = (torch.randn(B, C, generator=g) * 4).requires_grad_(True)
    lab = torch.randint(0, C, (B,), generator=g)
    beta = torch.rand(B, 1, generator=g).requires_grad_(True)
    ref = F.cross_entropy(x, lab)
    if mode == 1:
        ref = ref - coef * (beta * (1 - beta)).mean()
    elif mode == 2:
        b = torch.clamp(beta, 1e-8, 1 - 1e-8)
        ref = ref + coef * (-(b * torch.log(b) + (1 - b) * torch.log(1 - b))).mean()
    ref = ref * scale
    ref.backward()
    xd, bd = x.detach().cuda().requires_grad_(True), beta.detach().cuda().requires_grad_(True)
    loss = _ops.FusionLossCEFn.apply(xd, bd if mode else None, lab.cuda(), mode, coef, scale)
    (loss * 3.0).backward()
    assert abs(float(loss) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    assert (xd.grad.cpu() / 3.0 - x.grad).abs().max() <= 2e-6 * max(1.0, x.grad.abs().max().item())
    if mode:
        assert (bd.grad.cpu() / 3.0 - beta.grad).abs().max() <= 1e-5 * max(1e-3, beta.grad.abs().max().item())
    if mode == 1 and coef == 0.01 and scale == 1.0:
        l2 = fusion_step_loss_single_label(x.detach().cuda(), beta.detach().cuda(), lab.cuda())
        assert abs(float(l2) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
        cpu = fusion_step_loss_single_label(x.detach(), beta.detach(), lab)          # the CPU formula the gloo tests use
        assert abs(float(cpu) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))


def test_fusion_loss_ce_label_out_of_range_is_nan(lib):
    """torch raises for a class index outside [0, C); a kernel cannot, so the loss is poisoned instead of reading out of bounds"""
    from hri_emo_amd import _ops
    x = torch.randn(4, 3).cuda()
    loss = _ops.FusionLossCEFn.apply(x, None, torch.tensor([0, 2, 3, 1]).cuda(), 0, 0.0, 1.0)
    assert torch.isnan(loss).item()


def test_fusion_loss_ce_skips_ignore_index_like_torch(lib):
    """nn.CrossEntropyLoss() (the IEMOCAP trainer's criterion, train_fusion_seq_level_decoder.py:413-414) skips samples labelled
    -100 (ignore_index): no loss term, zero gradient, mean over the labelled samples only; every sample ignored gives NaN"""
    from hri_emo_amd import _ops
    g = torch.Generator().manual_seed(11)
    for B, C in ((9, 4), (300, 6)):
        x = (torch.randn(B, C, generator=g) * 3).requires_grad_(True)
        lab = torch.randint(0, C, (B,), generator=g)
        lab[::3] = -100
        beta = torch.rand(B, 1, generator=g).requires_grad_(True)
        ref = F.cross_entropy(x, lab) - 0.01 * (beta * (1 - beta)).mean()
        ref.backward()
        xd, bd = x.detach().cuda().requires_grad_(True), beta.detach().cuda().requires_grad_(True)
        loss = _ops.FusionLossCEFn.apply(xd, bd, lab.cuda(), 1, 0.01, 1.0)
        loss.backward()
        assert abs(float(loss) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
        assert (xd.grad.cpu() - x.grad).abs().max() <= 2e-6 * max(1.0, x.grad.abs().max().item())
        assert (xd.grad.cpu()[::3] == 0).all()
        assert (bd.grad.cpu() - beta.grad).abs().max() <= 1e-5 * max(1e-3, beta.grad.abs().max().item())
    x = torch.randn(4, 3).cuda()
    assert torch.isnan(_ops.FusionLossCEFn.apply(x, None, torch.full((4,), -100).cuda(), 0, 0.0, 1.0)).item()


def test_batched_cast_and_copy_jobs_in_one_launch(lib):
    """hriemo_cast_copy_batch / hriemo_cast_f32_to_bf16_batch: a table of jobs passed through kernel arguments -- fp32 -> bf16
    casts (kind 0) and fp32 copies (kind 1) of different lengths incl. tails that are no multiple of 8 and more than 64 jobs --
    each destination exactly bf16(src) / src, nothing written past a job's end"""
    g = torch.Generator().manual_seed(3)
    sizes = [8, 24, 2048, 2056, 5000, 768 * 96, 17 * 8, 100000] + [16 * (j + 1) for j in range(70)]
    srcs = [torch.randn(n, generator=g).cuda() for n in sizes]
    kinds = [(j % 3 == 1) * 1 for j in range(len(sizes))]
    dsts = [torch.full((n + 8,), 7.0, dtype=torch.float32 if k else torch.bfloat16, device="cuda") for n, k in zip(sizes, kinds)]
    jobs = torch.tensor([(s.data_ptr(), d.data_ptr(), n, k) for s, d, n, k in zip(srcs, dsts, sizes, kinds)], dtype=torch.int64)
    lib.call("hriemo_cast_copy_batch", jobs.data_ptr(), len(sizes), ST())
    torch.cuda.synchronize()
    for s, d, n, k in zip(srcs, dsts, sizes, kinds):
        assert torch.equal(d[:n], s if k else s.bfloat16()), (n, k)
        assert bool((d[n:].float() == 7.0).all()), (n, k, "wrote past the end")
    # the 3-column form: casts only
    d2 = [torch.empty(n, dtype=torch.bfloat16, device="cuda") for n in sizes[:5]]
    jobs3 = torch.tensor([(s.data_ptr(), d.data_ptr(), n) for s, d, n in zip(srcs, d2, sizes)], dtype=torch.int64)
    lib.call("hriemo_cast_f32_to_bf16_batch", jobs3.data_ptr(), 5, ST())
    for s, d in zip(srcs, d2):
        assert torch.equal(d, s.bfloat16())


def test_expand_rows_with_fp32_twin_and_seed_bump(lib):
    """hriemo_expand_rows: out[b] = bf16(q) and, with a twin pointer, out32[b] = q (emotion_decoder.py:127);
    hriemo_seed_bump: the device seed word += 0x9E3779B97F4A7C15 mod 2^64, once per launch"""
    q = torch.randn(6, 768, generator=torch.Generator().manual_seed(4)).cuda()
    B = 5
    out, out32 = bf(B, 6, 768), f32(B, 6, 768)
    lib.call("hriemo_expand_rows", P(q), P(out), P(out32), B, 6 * 768, ST())
    assert torch.equal(out, q.bfloat16()[None].expand(B, -1, -1)) and torch.equal(out32, q[None].expand(B, -1, -1))
    out_b = bf(B, 6, 768)
    lib.call("hriemo_expand_rows", P(q), P(out_b), None, B, 6 * 768, ST())
    assert torch.equal(out_b, out)
    word = torch.tensor([0x7FFFFFFFFFFFFFF0], dtype=torch.int64, device="cuda")
    want = 0x7FFFFFFFFFFFFFF0
    for _ in range(3):
        lib.call("hriemo_seed_bump", P(word), ST())
        want = (want + 0x9E3779B97F4A7C15) & ((1 << 64) - 1)
    got = int(word.item()) & ((1 << 64) - 1)
    assert got == want, (hex(got), hex(want))


@pytest.mark.parametrize("B,La,Lt,d,masked,twin", [c for c in GATE_CASES if c[3] <= 1024] + [(64, 400, 128, 768, True, True)])
def test_ln_pool_pair_launch_equals_the_two_single_launches(lib, B, La, Lt, d, masked, twin):
    """hriemo_ln_pool_fwd_pair / hriemo_ln_pool_bwd_pair: both modalities of the gate (models/beta_gate_tacfn.py:79-84 and its
    backward) from ONE launch instead of two on two streams.  The blocks run the single launches' code on the same operands, so
    every output -- normalised rows, mean / rstd, pooled partial sums; dX, and dgamma / dbeta through the same column reduce --
    must hold the same bits (bit-exact, like the rest of the integer-free but order-fixed row kernels)."""
    assert lib.lib().hriemo_ln_pool_pair_supported(d) == 1 and lib.lib().hriemo_ln_pool_pair_supported(2048) == 0
    g = torch.Generator().manual_seed(B * 131 + La)
    L = La if La == Lt else Lt
    if La < L:
        pytest.skip("audio shorter than text: the reference cannot fuse this either")

    def side(Lx, seed):
        gg = torch.Generator().manual_seed(seed)
        x32 = torch.randn(B, Lx, d, generator=gg).cuda()
        x = x32.bfloat16()
        m = ragged_mask(B, Lx, gg).cuda().to(torch.uint8) if masked else None
        return x, (x32 if twin else None), m, (1 + 0.1 * torch.randn(d, generator=gg)).cuda(), (0.1 * torch.randn(d, generator=gg)).cuda()

    xa, xa32, ma, ga, ba = side(La, 1)
    xt, xt32, mt, gt, bt = side(Lt, 2)
    nca, nct = lib.lib().hriemo_pool_chunks(La), lib.lib().hriemo_pool_chunks(Lt)

    def fwd_outs():
        return dict(An=bf(B, L, d).zero_(), Tn=bf(B, L, d).zero_(), mean_a=f32(B * La), rstd_a=f32(B * La), mean_t=f32(B * Lt), rstd_t=f32(B * Lt),
                    pa=f32(B, nca, d), pt=f32(B, nct, d))

    s, p = fwd_outs(), fwd_outs()
    lib.call("hriemo_ln_pool_fwd", P(xa), P(xa32), P(ma), P(ga), P(ba), P(s["An"]), P(s["mean_a"]), P(s["rstd_a"]), P(s["pa"]), B, La, L, d, 1e-5, ST())
    lib.call("hriemo_ln_pool_fwd", P(xt), P(xt32), P(mt), P(gt), P(bt), P(s["Tn"]), P(s["mean_t"]), P(s["rstd_t"]), P(s["pt"]), B, Lt, L, d, 1e-5, ST())
    lib.call("hriemo_ln_pool_fwd_pair", P(xa), P(xa32), P(ma), P(ga), P(ba), P(p["An"]), P(p["mean_a"]), P(p["rstd_a"]), P(p["pa"]), La,
             P(xt), P(xt32), P(mt), P(gt), P(bt), P(p["Tn"]), P(p["mean_t"]), P(p["rstd_t"]), P(p["pt"]), Lt, B, L, d, 1e-5, ST())
    torch.cuda.synchronize()
    for k in s:
        assert torch.equal(s[k], p[k]), ("forward", k)
        assert torch.isfinite(s[k].float()).all(), k

    dH = (0.1 * torch.randn(B, L, d, generator=g)).cuda().bfloat16()
    w = torch.rand(B, d, generator=g).cuda()
    da, dt = (0.01 * torch.randn(B, d, generator=g)).cuda(), (0.01 * torch.randn(B, d, generator=g)).cuda()
    nba, nbt = lib.lib().hriemo_ln_pool_bwd_workspace_bytes(B, La, d), lib.lib().hriemo_ln_pool_bwd_workspace_bytes(B, Lt, d)

    def bwd_outs():
        return dict(dxa=bf(B, La, d), dxt=bf(B, Lt, d), dga=f32(d).fill_(0.5), dba=f32(d).fill_(0.25), dgt=f32(d).fill_(-0.5), dbt=f32(d).fill_(1.0))

    for acc in (0, 1):
        s, p = bwd_outs(), bwd_outs()
        wsa, wst = f32(nba // 4 + 16), f32(nbt // 4 + 16)
        fo = fwd_outs()                     # statistics of the forward (recomputed by the single launches, shared by both paths)
        lib.call("hriemo_ln_pool_fwd", P(xa), P(xa32), P(ma), P(ga), P(ba), P(fo["An"]), P(fo["mean_a"]), P(fo["rstd_a"]), P(fo["pa"]), B, La, L, d, 1e-5, ST())
        lib.call("hriemo_ln_pool_fwd", P(xt), P(xt32), P(mt), P(gt), P(bt), P(fo["Tn"]), P(fo["mean_t"]), P(fo["rstd_t"]), P(fo["pt"]), B, Lt, L, d, 1e-5, ST())
        lib.call("hriemo_ln_pool_bwd", P(dH), L, P(w), 1, P(da), P(ma), P(xa), P(xa32), P(ga), P(fo["mean_a"]), P(fo["rstd_a"]), P(s["dxa"]),
                 P(s["dga"]), P(s["dba"]), acc, B, La, d, P(wsa), ST())
        lib.call("hriemo_ln_pool_bwd", P(dH), L, P(w), 0, P(dt), P(mt), P(xt), P(xt32), P(gt), P(fo["mean_t"]), P(fo["rstd_t"]), P(s["dxt"]),
                 P(s["dgt"]), P(s["dbt"]), acc, B, Lt, d, P(wst), ST())
        wsa2, wst2 = f32(nba // 4 + 16), f32(nbt // 4 + 16)
        lib.call("hriemo_ln_pool_bwd_pair", P(dH), L, P(w),
                 P(da), P(ma), P(xa), P(xa32), P(ga), P(fo["mean_a"]), P(fo["rstd_a"]), P(p["dxa"]), P(p["dga"]), P(p["dba"]), La, P(wsa2),
                 P(dt), P(mt), P(xt), P(xt32), P(gt), P(fo["mean_t"]), P(fo["rstd_t"]), P(p["dxt"]), P(p["dgt"]), P(p["dbt"]), Lt, P(wst2),
                 acc, B, d, ST())
        torch.cuda.synchronize()
        for k in s:
            assert torch.equal(s[k], p[k]), ("backward", acc, k)
            assert torch.isfinite(s[k].float()).all(), k
    # partial sums left in the workspaces when no dgamma / dbeta is given (the launch-boundary reduce finishes them): same words
    nch_a, nch_t = lib.lib().hriemo_ln_pool_bwd_chunks(La), lib.lib().hriemo_ln_pool_bwd_chunks(Lt)
    wsa3, wst3 = f32(nba // 4 + 16).zero_(), f32(nbt // 4 + 16).zero_()
    q = bwd_outs()
    lib.call("hriemo_ln_pool_bwd_pair", P(dH), L, P(w),
             P(da), P(ma), P(xa), P(xa32), P(ga), P(fo["mean_a"]), P(fo["rstd_a"]), P(q["dxa"]), None, None, La, P(wsa3),
             P(dt), P(mt), P(xt), P(xt32), P(gt), P(fo["mean_t"]), P(fo["rstd_t"]), P(q["dxt"]), None, None, Lt, P(wst3), 1, B, d, ST())
    torch.cuda.synchronize()
    assert torch.equal(wsa3[:B * nch_a * 2 * d], wsa2[:B * nch_a * 2 * d]) and torch.equal(wst3[:B * nch_t * 2 * d], wst2[:B * nch_t * 2 * d])
