"""GPU suite of the packed (varlen) path, SURVEY 8(f) rank 4: the encoder on the valid rows only (hriemo_attn_*_varlen with
cu_seqlens, GEMM / LayerNorm / FFN on [N_valid, d]) must return what the padded path returns on every valid row -- kernels first
(bit for bit), then the modules on the reference's ragged golden fixtures and a training step's gradients."""
import math

import pytest
import torch

from conftest import load_golden
from oracle import hri_emo_oracle as O          # the checker (tests only)

pytestmark = pytest.mark.gpu


@pytest.fixture()
def H():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import hri_emo_amd
    yield hri_emo_amd
    hri_emo_amd.set_varlen(False)


def cu(t):
    return None if t is None else t.cuda()


@pytest.mark.parametrize("B,H_,Lq,Lk,hd,p", [(5, 8, 100, 40, 96, 0.1), (4, 8, 128, 128, 96, 0.1), (6, 4, 333, 100, 64, 0.0), (3, 8, 50, 400, 96, 0.1),
                                              (7, 2, 70, 70, 32, 0.2), (3, 3, 40, 90, 128, 0.1), (4, 8, 6, 77, 16, 0.1)])
def test_attention_varlen_equals_padded_on_valid_rows(H, B, H_, Lq, Lk, hd, p):
    from hri_emo_amd import _ops as ops
    g = torch.Generator().manual_seed(Lq * 7 + Lk)
    d = H_ * hd
    lq = torch.randint(1, Lq + 1, (B,), generator=g); lk = torch.randint(1, Lk + 1, (B,), generator=g)
    lq[0], lk[0] = Lq, Lk                                   # the longest sample defines the padded shape
    if B > 2:
        lq[1], lk[1] = 1, 1                                 # and a one-row sample
    q = (torch.randn(B, Lq, d, generator=g) * 1.5).bfloat16().cuda()
    kv = torch.randn(B, Lk, 2 * d, generator=g).bfloat16().cuda()
    do = torch.randn(B, Lq, d, generator=g).bfloat16().cuda()
    vq = (torch.arange(Lq)[None] < lq[:, None]).cuda(); vk = (torch.arange(Lk)[None] < lk[:, None]).cuda()
    do = do * vq[:, :, None]                              # PAD query rows carry no gradient in the model (nothing reads them)
    kpm = (~vk).view(torch.uint8)
    seed, site, boff = 987654321, 12, 3
    # padded reference run (our own padded kernels with a key padding mask)
    q2, kv2, do2 = q.view(B * Lq, d), kv.view(B * Lk, 2 * d), do.view(B * Lq, d)
    o, lse, mb = ops.attn_fwd(q2, kv2[:, :d], kv2[:, d:], B, H_, Lq, Lk, hd, kpm, p, seed, site, boff, want_bits=True)
    dq = torch.empty_like(q2); dkv = torch.empty_like(kv2)
    ops.attn_bwd(q2, kv2[:, :d], kv2[:, d:], o, do2, dq, dkv[:, :d], dkv[:, d:], lse, B, H_, Lq, Lk, hd, kpm, p, seed, site, boff, mask_bits=mb)
    # packed run
    iq, ik = vq.reshape(-1).nonzero().reshape(-1), vk.reshape(-1).nonzero().reshape(-1)
    cq = torch.zeros(B + 1, dtype=torch.int32); cq[1:] = torch.cumsum(lq, 0)
    ck = torch.zeros(B + 1, dtype=torch.int32); ck[1:] = torch.cumsum(lk, 0)
    cq, ck = cq.cuda(), ck.cuda()
    qp, kvp, dop = q2.index_select(0, iq).contiguous(), kv2.index_select(0, ik).contiguous(), do2.index_select(0, iq).contiguous()
    for use_bits in (True, False):
        op, lsep, mbp = ops.attn_fwd(qp, kvp[:, :d], kvp[:, d:], B, H_, Lq, Lk, hd, None, p, seed, site, boff, want_bits=True, cu=(cq, ck))
        dqp = torch.empty_like(qp); dkvp = torch.empty_like(kvp)
        ops.attn_bwd(qp, kvp[:, :d], kvp[:, d:], op, dop, dqp, dkvp[:, :d], dkvp[:, d:], lsep, B, H_, Lq, Lk, hd, None, p, seed, site, boff,
                     mask_bits=mbp if use_bits else None, cu=(cq, ck))
        assert torch.equal(op, o.index_select(0, iq)), "O"
        vq3 = vq[:, None, :].expand(B, H_, Lq)
        assert torch.equal(lsep[vq3], lse[vq3]), "lse"
        assert torch.equal(dqp, dq.index_select(0, iq)), ("dQ", use_bits)
        assert torch.equal(dkvp, dkv.index_select(0, ik)), ("dK|dV", use_bits)
    # the padded run leaves exact zeros in dK / dV of PAD keys, nothing else to compare there
    assert float(dkv.float()[(~vk).reshape(-1)].abs().max() if (~vk).any() else 0.0) == 0.0


def test_attention_varlen_bias_colsums_match(H):
    """the in-projection bias gradients come from the attention kernels' column-sum partials: packed == padded"""
    from hri_emo_amd import _ops as ops
    B, H_, Lq, Lk, hd, p = 6, 8, 200, 90, 96, 0.1
    d = H_ * hd
    g = torch.Generator().manual_seed(5)
    lq = torch.randint(1, Lq + 1, (B,), generator=g); lk = torch.randint(1, Lk + 1, (B,), generator=g)
    lq[0], lk[0] = Lq, Lk
    q2 = torch.randn(B * Lq, d, generator=g).bfloat16().cuda(); kv2 = torch.randn(B * Lk, 2 * d, generator=g).bfloat16().cuda()
    do2 = torch.randn(B * Lq, d, generator=g).bfloat16().cuda()
    vq = (torch.arange(Lq)[None] < lq[:, None]).cuda(); vk = (torch.arange(Lk)[None] < lk[:, None]).cuda()
    do2 = do2 * vq.reshape(-1, 1)                         # upstream gradient of PAD query rows is zero in the model
    iq, ik = vq.reshape(-1).nonzero().reshape(-1), vk.reshape(-1).nonzero().reshape(-1)
    cq = torch.zeros(B + 1, dtype=torch.int32); cq[1:] = torch.cumsum(lq, 0)
    ck = torch.zeros(B + 1, dtype=torch.int32); ck[1:] = torch.cumsum(lk, 0)
    res = []
    for packed in (False, True):
        if packed:
            qq, kk, dd, kpm, cuu = q2.index_select(0, iq).contiguous(), kv2.index_select(0, ik).contiguous(), do2.index_select(0, iq).contiguous(), None, (cq.cuda(), ck.cuda())
        else:
            qq, kk, dd, kpm, cuu = q2, kv2, do2, (~vk).view(torch.uint8), None
        o, lse = ops.attn_fwd(qq, kk[:, :d], kk[:, d:], B, H_, Lq, Lk, hd, kpm, p, 11, 3, 0, cu=cuu)
        dq = torch.empty_like(qq); dkv = torch.empty_like(kk)
        bq = torch.zeros(d, device="cuda"); bkv = torch.zeros(2 * d, device="cuda")
        ops.attn_bwd(qq, kk[:, :d], kk[:, d:], o, dd, dq, dkv[:, :d], dkv[:, d:], lse, B, H_, Lq, Lk, hd, kpm, p, 11, 3, 0, bias_grad=(bq, bkv), cu=cuu)
        res.append((bq.cpu(), bkv.cpu(), dq.float().sum(0).cpu(), dkv.float().sum(0).cpu()))
    (bq0, bkv0, sq0, skv0), (bq1, bkv1, sq1, skv1) = res
    tol = 2e-3
    assert float((bq1 - bq0).abs().max()) <= tol * float(bq0.abs().max())
    assert float((bkv1 - bkv0).abs().max()) <= tol * float(bkv0.abs().max())
    assert float((bq1 - sq1).abs().max()) <= 2e-2 * float(sq1.abs().max())       # partials are sums of the unrounded values


def _fusion(H, d, ne, p=0.1):
    return O.closed_form_init_(H.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=p)).cuda()


@pytest.mark.parametrize("name,d,ne", [("cfg1_eval_ragged", 128, 4), ("hd96_eval_ragged", 768, 6)])
def test_fusion_eval_varlen_equals_padded_and_golden(H, name, d, ne):
    g = load_golden(name)
    m = _fusion(H, d, ne).eval()
    args = (cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]))
    with torch.no_grad():
        H.set_varlen(False)
        ref = m(*args)
        H.set_varlen(True)
        got = m(*args)
    from hri_emo_amd import _ops
    assert _ops.seq_plan(args[2], *args[2].shape) is not None, "the fixture's masks are prefix masks: the packed path must have run"
    valid = 1.0 - g["mask_a"].float().mean().item()
    for a, b, what in zip(got, ref, ("logits", "beta", "z")):
        assert float((a.float() - b.float()).abs().max()) <= 1e-5 * max(1.0, float(b.float().abs().max())), what
    for a, what in zip(got, ("logits", "beta", "z")):
        r = g[what]
        assert float((a.float().cpu() - r).abs().max()) <= 5e-3 * max(1.0, float(r.abs().max())), what
    print(f"{name}: packed == padded; valid audio fraction {valid:.2f}")


def test_varlen_falls_back_on_masks_that_are_not_prefixes(H):
    from hri_emo_amd import _ops
    g = load_golden("cfg1_eval_ragged")
    m = _fusion(H, 128, 4).eval()
    ma = g["mask_a"].clone()
    ma[0, 3] = True                                   # a hole inside the valid prefix
    assert _ops.seq_plan(ma.cuda(), *ma.shape) is None
    with torch.no_grad():
        H.set_varlen(False)
        ref = m(cu(g["h_a"]), cu(g["h_t"]), cu(ma), cu(g["mask_t"]))
        H.set_varlen(True)
        got = m(cu(g["h_a"]), cu(g["h_t"]), cu(ma), cu(g["mask_t"]))
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


def test_train_step_gradients_varlen_vs_padded(H):
    """dropout 0: every parameter gradient of the packed step equals the padded step's to fp32 summation-order tolerance (the
    weight-gradient GEMMs contract over N_valid rows instead of B*L rows whose PAD entries contribute exact zeros)"""
    from hri_emo_amd.train import fusion_step_loss
    torch.manual_seed(3)
    kw = dict(d_model=256, num_emotions=5, n_heads=8, dropout=0.0)
    m = H.FusionWithEmotionDecoder(**kw).cuda().train()
    g = torch.Generator().manual_seed(4)
    B, Ta, Tt, d = 6, 150, 60, 256
    h_a, h_t = torch.randn(B, Ta, d, generator=g).cuda(), torch.randn(B, Tt, d, generator=g).cuda()
    la = torch.randint(40, Ta + 1, (B,), generator=g); lt = torch.randint(10, Tt + 1, (B,), generator=g)
    la[0], lt[0] = Ta, Tt
    m_a, m_t = (torch.arange(Ta)[None] >= la[:, None]).cuda(), (torch.arange(Tt)[None] >= lt[:, None]).cuda()
    y = (torch.rand(B, 5, generator=g) < 0.3).float().cuda()
    grads = []
    for packed in (False, True):
        H.set_varlen(packed)
        m.zero_grad(set_to_none=True)
        logits, beta, z = m(h_a, h_t, m_a, m_t)
        loss = fusion_step_loss(logits, beta, y)
        loss.backward()
        grads.append(({n: p.grad.detach().float().clone() for n, p in m.named_parameters()}, float(loss)))
    (g0, l0), (g1, l1) = grads
    assert abs(l0 - l1) <= 1e-5 * max(1.0, abs(l0))
    worst = 0.0
    for n in g0:
        rel = float((g1[n] - g0[n]).norm() / g0[n].norm().clamp_min(1e-20))
        worst = max(worst, rel)
        assert rel <= 1e-5, (n, rel)          # measured 1.2e-7 (fp32 summation order of the weight-gradient GEMMs)
    print(f"packed vs padded gradients: worst relative L2 difference {worst:.2e}; valid fraction audio {float(la.sum()) / (B * Ta):.2f} text {float(lt.sum()) / (B * Tt):.2f}")


def test_train_step_with_dropout_packed_equals_padded(H):
    """dropout on: the attention masks are keyed by position within the sequence and the LayerNorm masks by the row of the PADDED
    layout (hriemo_add_ln_*_rows), so a packed step draws exactly the masks of the padded step from the same seed: loss and every
    parameter gradient agree to fp32 summation-order tolerance, as with dropout off"""
    from hri_emo_amd.train import fusion_step_loss
    torch.manual_seed(3)
    m = H.FusionWithEmotionDecoder(d_model=256, num_emotions=5, n_heads=8, dropout=0.1).cuda().train()
    g = torch.Generator().manual_seed(4)
    B, Ta, Tt, d = 8, 120, 48, 256
    h_a, h_t = torch.randn(B, Ta, d, generator=g).cuda(), torch.randn(B, Tt, d, generator=g).cuda()
    la = torch.randint(30, Ta + 1, (B,), generator=g); lt = torch.randint(10, Tt + 1, (B,), generator=g)
    la[0], lt[0] = Ta, Tt
    m_a, m_t = (torch.arange(Ta)[None] >= la[:, None]).cuda(), (torch.arange(Tt)[None] >= lt[:, None]).cuda()
    y = (torch.rand(B, 5, generator=g) < 0.3).float().cuda()
    runs = []
    for packed in (False, True):
        H.set_varlen(packed)
        m.zero_grad(set_to_none=True)
        torch.manual_seed(77)                      # the step's dropout seed comes from torch's generator
        logits, beta, z = m(h_a, h_t, m_a, m_t)
        loss = fusion_step_loss(logits, beta, y)
        loss.backward()
        runs.append((float(loss), {n: p.grad.detach().float().clone() for n, p in m.named_parameters()}))
    (l0, g0), (l1, g1) = runs
    assert abs(l0 - l1) <= 1e-5 * max(1.0, abs(l0)), (l0, l1)
    worst = max(float((g1[n] - g0[n]).norm() / g0[n].norm().clamp_min(1e-20)) for n in g0)
    assert worst <= 1e-4, worst
    torch.manual_seed(78)
    H.set_varlen(True)
    l2 = float(fusion_step_loss(*m(h_a, h_t, m_a, m_t)[:2], y))
    assert abs(l2 - l1) > 1e-6                     # and another seed is another draw (the dropout is really on)


def test_pack_unpack_rows_kernels_with_device_side_lengths(H):
    """hriemo_pack_rows / hriemo_unpack_rows through the C-ABI: the lengths are read from device memory (cu_seqlens), bucket
    padding rows come out as zeros, row_index is the padded row of every packed row, unpack zero-fills the PAD positions; the
    bf16 tensor and its fp32 twin travel in one launch"""
    from hri_emo_amd import _lib, _ops
    g = torch.Generator().manual_seed(5)
    B, L, d, n_rows = 5, 37, 72, 160
    lens = torch.tensor([37, 1, 20, 33, 9])
    cu_ = torch.zeros(B + 2, dtype=torch.int32)
    cu_[1:B + 1] = torch.cumsum(lens, 0)
    cu_[B + 1] = n_rows
    cu_ = cu_.cuda()
    x32 = torch.randn(B, L, d, generator=g).cuda()
    x16 = x32.bfloat16()
    N = int(lens.sum())
    idx = torch.cat([b * L + torch.arange(int(lens[b])) for b in range(B)]).cuda()
    st = _ops._stream()
    for with16, with32 in ((True, True), (True, False), (False, True)):
        p16 = torch.full((n_rows, d), 7.0, dtype=torch.bfloat16, device="cuda") if with16 else None
        p32 = torch.full((n_rows, d), 7.0, device="cuda") if with32 else None
        rows = torch.full((n_rows,), -1, dtype=torch.int64, device="cuda")
        _lib.call("hriemo_pack_rows", _ops._p(x16 if with16 else None), _ops._p(x32 if with32 else None), _ops._p(cu_), B, L, d, n_rows,
                  _ops._p(p16), _ops._p(p32), _ops._p(rows), st)
        assert torch.equal(rows[:N], idx) and torch.equal(rows[N:], B * L + torch.arange(n_rows - N, device="cuda"))
        if with16:
            assert torch.equal(p16[:N], x16.view(B * L, d)[idx]) and float(p16[N:].float().abs().sum()) == 0.0
        if with32:
            assert torch.equal(p32[:N], x32.view(B * L, d)[idx]) and float(p32[N:].abs().sum()) == 0.0
        y16 = torch.full((B, L, d), 3.0, dtype=torch.bfloat16, device="cuda") if with16 else None
        y32 = torch.full((B, L, d), 3.0, device="cuda") if with32 else None
        _lib.call("hriemo_unpack_rows", _ops._p(p16), _ops._p(p32), _ops._p(cu_), B, L, d, _ops._p(y16), _ops._p(y32), st)
        valid = (torch.arange(L)[None] < lens[:, None]).cuda()
        if with16:
            assert torch.equal(y16[valid], x16[valid]) and float(y16[~valid].float().abs().sum()) == 0.0
        if with32:
            assert torch.equal(y32[valid], x32[valid]) and float(y32[~valid].abs().sum()) == 0.0


def _ragged_batch(B, Ta, Tt, d, ne, seed, lo_a, lo_t):
    g = torch.Generator().manual_seed(seed)
    h_a, h_t = torch.randn(B, Ta, d, generator=g).cuda(), torch.randn(B, Tt, d, generator=g).cuda()
    la = torch.randint(lo_a, Ta + 1, (B,), generator=g); lt = torch.randint(lo_t, Tt + 1, (B,), generator=g)
    m_a, m_t = (torch.arange(Ta)[None] >= la[:, None]).cuda(), (torch.arange(Tt)[None] >= lt[:, None]).cuda()
    y = (torch.rand(B, ne, generator=g) < 0.3).float().cuda()
    return (h_a, h_t, m_a, m_t, y), (la.tolist(), lt.tolist())


def test_captured_packed_step_serves_every_batch(H):
    """One capture in packed mode, then batches with OTHER padding masks: the lengths are device data of the graph (cu_seqlens),
    the packed row count is rounded up to a bucket whose surplus rows form an all-zero extra sequence.  Every replay must give
    the loss and the parameter gradients of the padded eager step on the same batch (dropout 0) -- for a batch in the captured
    bucket, for batches that open new buckets (captured on the spot), with the lengths read from the masks or handed over."""
    from hri_emo_amd.dp import DataParallelStep
    from hri_emo_amd.train import fusion_step_loss
    torch.manual_seed(3)
    m = H.FusionWithEmotionDecoder(d_model=256, num_emotions=5, n_heads=8, dropout=0.0).cuda().train()
    B, Ta, Tt, d = 6, 150, 60, 256
    dp = DataParallelStep(m, fusion_step_loss, overlap=False)
    dp.set_global_batch(B)
    batches = [_ragged_batch(B, Ta, Tt, d, 5, s, lo_a, lo_t) for s, lo_a, lo_t in ((4, 40, 10), (5, 40, 10), (6, 120, 50), (7, 1, 1), (8, 150, 60))]
    ref = []
    H.set_varlen(False)
    for batch, _ in batches:                      # the padded eager step is the yardstick
        loss = float(dp.step(*batch))
        ref.append((loss, dp.buckets.flat.clone()))
    H.set_varlen(True)
    dp.capture(*batches[0][0])
    seen = set()
    for i, (batch, lens) in enumerate(batches):
        loss = float(dp.step(*batch, lengths=lens if i % 2 else None))
        torch.cuda.synchronize()
        seen.add(tuple(int(x) for x in (dp._pb["cu_a"][-1], dp._pb["cu_t"][-1])))
        assert abs(loss - ref[i][0]) <= 1e-5 * max(1.0, abs(ref[i][0])), (i, loss, ref[i][0])
        rel = float((dp.buckets.flat - ref[i][1]).norm() / ref[i][1].norm())
        assert rel <= 1e-5, (i, rel)
    assert len(dp._pb["graphs"]) == len(seen) >= 3            # full-length and very short batches fall into other buckets
    loss = float(dp.step(*batches[1][0]))                         # back to a bucket that exists: a replay, same numbers
    assert abs(loss - ref[1][0]) <= 1e-5 * max(1.0, abs(ref[1][0]))
    with pytest.raises(RuntimeError, match="suffix"):
        bad = batches[1][0][2].clone(); bad[0, 3] = True; bad[0, 4] = False
        dp.step(batches[1][0][0], batches[1][0][1], bad, batches[1][0][3], batches[1][0][4])
    dp.release_graph()


def test_packed_step_reads_the_lengths_of_a_batch_that_reuses_freed_mask_addresses(H):
    """`dp.step(*make_batch())` -- every batch dropped before the next one is built, so the new masks get the freed
    addresses (and `_version` 0) of the old ones.  With lengths=None the step must still read THIS batch's lengths, not a cached
    cu_seqlens of the previous batch: every step equals the padded eager step of its own batch."""
    from hri_emo_amd.dp import DataParallelStep
    from hri_emo_amd.train import fusion_step_loss
    torch.manual_seed(3)
    m = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=8, dropout=0.0).cuda().train()
    B, Ta, Tt, d = 4, 96, 40, 128
    dp = DataParallelStep(m, fusion_step_loss, overlap=False)
    dp.set_global_batch(B)
    specs = [(31, 30, 10), (32, 60, 20), (33, 30, 10), (34, 90, 35), (35, 10, 5)]
    ref = []
    H.set_varlen(False)
    for s_, lo_a, lo_t in specs:
        batch, _ = _ragged_batch(B, Ta, Tt, d, 4, s_, lo_a, lo_t)
        ref.append((float(dp.step(*batch)), dp.buckets.flat.clone()))
        del batch
    H.set_varlen(True)
    first, _ = _ragged_batch(B, Ta, Tt, d, 4, *specs[0])
    dp.capture(*first)
    del first
    ptrs = set()
    for i, (s_, lo_a, lo_t) in enumerate(specs):
        batch, _ = _ragged_batch(B, Ta, Tt, d, 4, s_, lo_a, lo_t)        # the previous batch is gone: its blocks are free to be reused
        ptrs.add((batch[2].data_ptr(), batch[3].data_ptr()))
        loss = float(dp.step(*batch))                                    # lengths=None: read from the masks
        torch.cuda.synchronize()
        assert abs(loss - ref[i][0]) <= 1e-5 * max(1.0, abs(ref[i][0])), (i, loss, ref[i][0])
        rel = float((dp.buckets.flat - ref[i][1]).norm() / ref[i][1].norm())
        assert rel <= 1e-5, (i, rel)
        del batch
    dp.release_graph()


def test_packed_bucket_graphs_are_bounded(H, monkeypatch):
    """one hipGraph per distinct (audio rows, text rows) bucket pair must not grow without bound: beyond the cap the
    least recently used graph is released, and a bucket that comes back is captured again with the same results"""
    from hri_emo_amd import dp as dpmod
    from hri_emo_amd.train import fusion_step_loss
    monkeypatch.setattr(dpmod, "_VARLEN_MAX_GRAPHS", 2)
    torch.manual_seed(3)
    m = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=8, dropout=0.0).cuda().train()
    B, Ta, Tt, d = 4, 96, 40, 128
    dp = dpmod.DataParallelStep(m, fusion_step_loss, overlap=False)
    dp.set_global_batch(B)
    batches = [_ragged_batch(B, Ta, Tt, d, 4, s_, lo_a, lo_t)[0] for s_, lo_a, lo_t in ((41, 20, 8), (42, 50, 20), (43, 90, 38))]
    H.set_varlen(True)
    dp.capture(*batches[0])
    first = float(dp.step(*batches[0]))
    g0 = dp.buckets.flat.clone()
    rng = torch.get_rng_state()
    for b in batches[1:]:
        dp.step(*b)                                        # two new buckets: the first one's graph has to go
    assert torch.equal(torch.get_rng_state(), rng)         # bucket captures inside step() leave torch's CPU generator alone
    assert len(dp._pb["graphs"]) == 2
    again = float(dp.step(*batches[0]))                    # captured afresh
    torch.cuda.synchronize()
    assert len(dp._pb["graphs"]) == 2 and again == first and torch.equal(dp.buckets.flat, g0)
    dp.release_graph()


def test_captured_packed_step_with_dropout_is_deterministic_per_seed(H):
    """dropout on, packed bucket graph: replays from the same seed word are bit-identical (bucket padding rows carry zeros and
    zero gradients, so nothing of them leaks into a sum), another seed gives another loss"""
    from hri_emo_amd import _ops
    from hri_emo_amd.dp import DataParallelStep
    from hri_emo_amd.train import fusion_step_loss
    torch.manual_seed(3)
    m = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=8, dropout=0.1).cuda().train()
    B, Ta, Tt, d = 4, 64, 32, 128
    (h_a, h_t, m_a, m_t, y), _ = _ragged_batch(B, Ta, Tt, d, 4, 9, 10, 5)
    h_a, h_t = h_a.bfloat16(), h_t.bfloat16()
    H.set_varlen(True)
    dp = DataParallelStep(m, fusion_step_loss, overlap=False)
    dp.set_global_batch(B)
    dp.step(h_a, h_t, m_a, m_t, y)
    dp.capture(h_a, h_t, m_a, m_t, y)
    sw = _ops.seed_word(h_a.device)
    outs = []
    for seed in (123, 123, 124):
        sw.fill_(seed)
        loss = dp.step(h_a, h_t, m_a, m_t, y)
        torch.cuda.synchronize()
        outs.append((float(loss), dp.buckets.flat.clone()))
    assert math.isfinite(outs[0][0]) and bool(torch.isfinite(outs[0][1]).all())
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
    assert outs[2][0] != outs[0][0]
    dp.release_graph()


def test_packed_training_loop_end_to_end(H):
    """The loop INTEGRATION.md documents: fixed-shape collate (pad_to) -> DevicePrefetcher with host-side lengths -> one packed
    capture -> step + fused optimizer per batch, every batch another length pattern.  Dropout 0: the loss trajectory must follow
    the same loop run eagerly on the padded path (same initial weights, same data, same optimizer)."""
    import copy
    from hri_emo_amd import data
    from hri_emo_amd.dp import DataParallelStep
    from hri_emo_amd.optim import FusedClipAdamW
    from hri_emo_amd.train import fusion_step_loss
    g = torch.Generator().manual_seed(21)
    d, ne, B, Ta, Tt = 128, 4, 4, 48, 24
    samples = []
    for _ in range(6 * B):
        la, lt = int(torch.randint(5, Ta + 1, (1,), generator=g)), int(torch.randint(3, Tt + 1, (1,), generator=g))
        samples.append((torch.randn(la, d, generator=g), torch.zeros(la, dtype=torch.bool), torch.randn(lt, d, generator=g),
                        torch.zeros(lt, dtype=torch.bool), (torch.rand(ne, generator=g) < 0.3).float()))
    loader = [data.collate_seq_batch(samples[i:i + B], pad_to=(Ta, Tt)) for i in range(0, len(samples), B)]
    torch.manual_seed(3)
    m0 = H.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=0.0).cuda().train()
    traj = []
    for packed in (False, True):
        H.set_varlen(packed)
        m = copy.deepcopy(m0)
        dp = DataParallelStep(m, fusion_step_loss, overlap=False)
        dp.set_global_batch(B)
        opt = FusedClipAdamW(dp.buckets, lr=1e-3, weight_decay=1e-2, max_norm=5.0)
        batches = data.DevicePrefetcher(loader, "cuda", convert=lambda t: (t[0], t[2], t[1], t[3], t[4]), mask_slots=(2, 3))
        losses = []
        for i, (h_a, h_t, m_a, m_t, y, lens) in enumerate(batches):
            assert h_a.shape == (B, Ta, d) and h_t.shape == (B, Tt, d) and len(lens[0]) == B
            if packed and i == 0:
                dp.capture(h_a, h_t, m_a, m_t, y, lengths=lens)
            losses.append(float(dp.step(h_a, h_t, m_a, m_t, y, lengths=lens) if packed else dp.step(h_a, h_t, m_a, m_t, y)))
            opt.step()
        if packed:
            assert len(dp._pb["graphs"]) >= 2          # the six batches do not all fall into one bucket
            dp.release_graph()
        traj.append(losses)
    for a, b in zip(*traj):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), traj      # bf16 path, weights diverge slowly over the six updates
    assert traj[0][-1] != traj[0][0]


@pytest.mark.parametrize("case", ["all_full", "all_one", "single_utterance", "exact_bucket_multiple"])
def test_captured_packed_step_edge_length_patterns(H, case):
    """Bucket arithmetic at its edges: every sequence full length (the surplus sequence sits beyond B*L rows), every sequence one
    position long, a batch of one utterance, a valid row count that is an exact multiple of the bucket size (the bucket must still
    add at least one surplus row).  Loss and gradients = the padded eager step's."""
    from hri_emo_amd.dp import DataParallelStep
    from hri_emo_amd.train import fusion_step_loss
    torch.manual_seed(5)
    d, ne = 128, 4
    B, Ta, Tt = (1, 40, 16) if case == "single_utterance" else (4, 64, 32)
    m = H.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=0.0).cuda().train()
    g = torch.Generator().manual_seed(6)
    h_a, h_t = torch.randn(B, Ta, d, generator=g).cuda(), torch.randn(B, Tt, d, generator=g).cuda()
    y = (torch.rand(B, ne, generator=g) < 0.3).float().cuda()
    if case == "all_full":
        la, lt = [Ta] * B, [Tt] * B
    elif case == "all_one":
        la, lt = [1] * B, [1] * B
    elif case == "single_utterance":
        la, lt = [23], [9]
    else:                      # bucket sizes are max(8, B*L // 64 // 8 * 8) = 8 rows here: 4 x 16 = 64 audio rows, 4 x 8 = 32 text rows
        la, lt = [16] * B, [8] * B
    m_a = (torch.arange(Ta)[None] >= torch.tensor(la)[:, None]).cuda()
    m_t = (torch.arange(Tt)[None] >= torch.tensor(lt)[:, None]).cuda()
    dp = DataParallelStep(m, fusion_step_loss, overlap=False)
    dp.set_global_batch(B)
    H.set_varlen(False)
    ref_loss = float(dp.step(h_a, h_t, m_a, m_t, y))
    ref = dp.buckets.flat.clone()
    H.set_varlen(True)
    dp.capture(h_a, h_t, m_a, m_t, y, lengths=(la, lt))
    for _ in range(2):
        loss = float(dp.step(h_a, h_t, m_a, m_t, y, lengths=(la, lt)))
    torch.cuda.synchronize()
    rows_a, rows_t = int(dp._pb["cu_a"][-1]), int(dp._pb["cu_t"][-1])
    assert rows_a > sum(la) and rows_t > sum(lt) and rows_a - sum(la) <= Ta and rows_t - sum(lt) <= Tt, (rows_a, rows_t)
    assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (case, loss, ref_loss)
    rel = float((dp.buckets.flat - ref).norm() / ref.norm())
    assert rel <= 1e-5, (case, rel)
    dp.release_graph()
