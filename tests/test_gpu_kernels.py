"""GPU suite, kernel level: each C-ABI entry point against plain torch fp32 math on the same inputs.
GEMMs use small-integer operands so every layout/tail/epilogue case must match EXACTLY (a transposed or
permuted fragment cannot hide); attention / LayerNorm kernels are compared in fp32 with the tolerance a
bf16 result allows (stated per test)."""
import math

import numpy as np
import pytest
import torch

import hashrng

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import hri_emo_amd  # noqa: F401
    from hri_emo_amd import _ops
    return _ops


def ints(shape, lo=-3, hi=4, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).float()


GEMM_SHAPES = [(256, 256, 128), (200, 136, 96), (128, 384, 64), (1024, 768, 768), (64, 256, 3072), (37 * 8, 8, 32),
               (130, 2304, 768)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_nt_exact(ops, M, N, K):
    A, W, b = ints((M, K), seed=1), ints((N, K), seed=2), ints((N,), seed=3)
    ref = A @ W.t() + b
    y = ops.linear_fwd(A.cuda().bfloat16(), W.cuda().bfloat16(), b.cuda())
    assert torch.equal(y.float().cpu(), ref.bfloat16().float())
    yr = ops.linear_fwd(A.cuda().bfloat16(), W.cuda().bfloat16(), b.cuda(), relu=True)
    assert torch.equal(yr.float().cpu(), ref.clamp(min=0).bfloat16().float())
    yf = ops.linear_fwd(A.cuda().bfloat16(), W.cuda().bfloat16(), b.cuda(), out_f32=True)
    assert torch.equal(yf.cpu(), ref)


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_nn_exact_with_epilogues(ops, M, N, K):
    # dX[M,K] = dY[M,N] . W[N,K]
    dY, W = ints((M, N), seed=4), ints((N, K), seed=5)
    aux = ints((M, K), seed=6)
    ref = dY @ W
    dx = ops.linear_dx(dY.cuda().bfloat16(), W.cuda().bfloat16())
    assert torch.equal(dx.float().cpu(), ref.bfloat16().float())
    dx2 = ops.linear_dx(dY.cuda().bfloat16(), W.cuda().bfloat16(), epi=2, aux=aux.cuda().bfloat16())
    assert torch.equal(dx2.float().cpu(), (ref * (aux > 0)).bfloat16().float())
    dx3 = ops.linear_dx(dY.cuda().bfloat16(), W.cuda().bfloat16(), epi=3, aux=aux.cuda().bfloat16())
    assert torch.equal(dx3.float().cpu(), (ref + aux).bfloat16().float())


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES + [(4096, 768, 768), (32, 128, 512)])
def test_gemm_tn_exact_splitk(ops, M, N, K):
    # dW[N,K] = dY[M,N]^T . X[M,K]  (reduction over M rows; split-K when M is large)
    dY, X = ints((M, N), -2, 3, seed=7), ints((M, K), -2, 3, seed=8)
    ref = dY.t() @ X
    out = torch.empty((N, K), dtype=torch.float32, device="cuda")
    ops.linear_dw(dY.cuda().bfloat16(), X.cuda().bfloat16(), out)
    assert torch.equal(out.cpu(), ref)
    # strided views: column slices of wider buffers, row slice of the output
    wide = torch.zeros((M, N + 16), dtype=torch.bfloat16, device="cuda")
    wide[:, 8:8 + N] = dY.cuda().bfloat16()
    big = torch.zeros((N + 8, K), dtype=torch.float32, device="cuda")
    ops.linear_dw(wide[:, 8:8 + N], X.cuda().bfloat16(), big[8:])
    assert torch.equal(big[8:].cpu(), ref) and float(big[:8].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(25600, 768, 768), (8192, 3072, 768), (25600, 768, 2304)])
def test_gemm_tn_splitk_weight_gradient_shapes(ops, M, N, K):
    """Weight-gradient shapes of cfg 2 (split-K slabs + reduce): exact on integers, with accumulate, repeated on the same stream
    and interleaved with a second stream (private workspaces); bit-identical from run to run on random data (fixed summation
    order of the slabs)."""
    dY, X = ints((M, N), -2, 3, seed=21), ints((M, K), -2, 3, seed=22)
    dYd, Xd = dY.cuda().bfloat16(), X.cuda().bfloat16()
    ref = dY.t().double() @ X.double()
    out = torch.zeros((N, K), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    out2 = torch.zeros((N, K), dtype=torch.float32, device="cuda")
    for rep in range(3):
        ops.linear_dw(dYd, Xd, out, accumulate=True)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.linear_dw(dYd, Xd, out2, accumulate=True)
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(out.double().cpu(), (rep + 1) * ref), rep
        assert torch.equal(out2.double().cpu(), (rep + 1) * ref), rep
    g = torch.Generator().manual_seed(5)
    A, B_ = torch.randn(M, N, generator=g).cuda().bfloat16(), torch.randn(M, K, generator=g).cuda().bfloat16()
    r1 = torch.empty((N, K), dtype=torch.float32, device="cuda")
    r2 = torch.empty((N, K), dtype=torch.float32, device="cuda")
    ops.linear_dw(A, B_, r1)
    ops.linear_dw(A, B_, r2)
    assert torch.equal(r1, r2)
    want = A.float().t() @ B_.float()
    assert float((r1 - want).abs().max()) <= 2e-3 * float(want.abs().max())


@pytest.mark.parametrize("M,N,K,split", [(25600, 2304, 768, 768), (8192, 2304, 768, 768), (96, 384, 128, 128), (4096, 768, 256, 512)])
def test_gemm_tn_split_output(ops, M, N, K, split):
    """hriemo_gemm_bf16_split: one weight-gradient GEMM whose result rows go to two matrices (the Q rows of one in_proj_weight and
    the K | V rows of another one, _ops.SharedProjFn).  Exact on integers against torch, with and without accumulate, into row
    slices of wider parameters-like buffers; the small shape takes the path without split-K (two launches inside the library)."""
    dY, X = ints((M, N), -2, 3, seed=31), ints((M, K), -2, 3, seed=32)
    ref = dY.t().double() @ X.double()
    dYd, Xd = dY.cuda().bfloat16(), X.cuda().bfloat16()
    pa = torch.zeros((N, K), dtype=torch.float32, device="cuda")          # parameter A: its rows [0, split) are written
    pb = torch.zeros((N, K), dtype=torch.float32, device="cuda")          # parameter B: its rows [split, N) are written
    for rep in range(2):
        ops.linear_dw_split(dYd, Xd, pa[:split], pb[split:], split, accumulate=True)
        assert torch.equal(pa[:split].double().cpu(), (rep + 1) * ref[:split])
        assert torch.equal(pb[split:].double().cpu(), (rep + 1) * ref[split:])
    assert float(pa[split:].abs().max()) == 0.0 and float(pb[:split].abs().max()) == 0.0
    ops.linear_dw_split(dYd, Xd, pa[:split], pb[split:], split, accumulate=False)
    assert torch.equal(pa[:split].double().cpu(), ref[:split]) and torch.equal(pb[split:].double().cpu(), ref[split:])


def test_gemm_group_tn_many_weight_gradients_in_one_launch(ops):
    """hriemo_gemm_bf16_group_tn: the decoder's / gate's weight-gradient GEMMs (short reductions, 64 ... 1000 rows; outputs from
    256 x 3072 down to 768 x 256, row slices of wider buffers, strided operands) as one grouped launch per 16 problems -- 19 problems
    here, i.e. two launches -- exact on integers against torch, accumulating into non-zero destinations."""
    from hri_emo_amd import _lib
    shapes = [(384, 2304, 768), (384, 768, 768), (384, 768, 2048), (384, 2048, 768), (64, 768, 256), (64, 256, 3072), (1000, 136, 200),
              (8, 64, 64), (384, 1536, 768)] * 2 + [(72, 8, 8)]
    keep, table, want = [], [], []
    for j, (K, M, N) in enumerate(shapes):                     # K reduction rows; result [M, N]
        dy = torch.zeros((K, M + 16), dtype=torch.bfloat16, device="cuda")
        dyv = ints((K, M), -2, 3, seed=50 + j)
        dy[:, 8:8 + M] = dyv.cuda().bfloat16()
        x = ints((K, N), -2, 3, seed=90 + j)
        xd = x.cuda().bfloat16()
        init = ints((M + 8, N), -1, 2, seed=130 + j)
        out = init.clone().cuda()
        keep += [dy, xd, out]
        a = dy[:, 8:8 + M]
        table.append((M, N, K, a.data_ptr(), a.stride(0), xd.data_ptr(), xd.stride(0), out[8:].data_ptr(), out.stride(0)))
        want.append((out, init, dyv.t().double() @ x.double()))
    host = torch.tensor(table, dtype=torch.int64)
    _lib.call("hriemo_gemm_bf16_group_tn", host.data_ptr(), len(table), 1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for out, init, ref in want:
        assert torch.equal(out[:8].cpu(), init[:8])                                  # rows outside the destination untouched
        assert torch.equal(out[8:].double().cpu(), init[8:].double() + ref)
    _lib.call("hriemo_gemm_bf16_group_tn", host.data_ptr(), len(table), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for out, init, ref in want:
        assert torch.equal(out[8:].double().cpu(), ref)                              # accumulate = 0 overwrites


@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 3, 4, 6, 7, 8])
@pytest.mark.parametrize("M,N,K", [(200, 136, 96), (1000, 768, 256), (25600, 3072, 768), (8192, 3072, 768), (384, 2048, 768)])
def test_gemm_masked_dx_with_column_sums(ops, cfg, M, N, K):
    """dX = (dY . W) * (aux > 0) with the column sums of the stored tile out of the same launch (FFN1 bias gradient): dX exact on
    integers, column sums exact (they are sums of the rounded, masked values), every tile configuration, ragged edges"""
    from hri_emo_amd import _lib
    L = _lib.lib()
    if cfg >= 0 and M * K > 4e7:
        pytest.skip("forced configurations on the small shapes only")
    L.hriemo_gemm_force_config(cfg)
    try:
        dY, W, aux = ints((M, K), -2, 3, seed=31), ints((K, N), -2, 3, seed=32), ints((M, N), -3, 4, seed=33)
        ref = (dY @ W) * (aux > 0)
        out = torch.full((N,), 5.0, device="cuda")
        dx = ops.linear_dx_masked_colsum(dY.cuda().bfloat16(), W.cuda().bfloat16(), aux.cuda().bfloat16(), out, True)
        assert torch.equal(dx.float().cpu(), ref.bfloat16().float()), (cfg, "dX")
        want = ref.bfloat16().float().double().sum(0) + 5.0
        assert float((out.double().cpu() - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max())), (cfg, "column sums")
    finally:
        L.hriemo_gemm_force_config(-1)


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8])
def test_gemm_every_tile_config_and_persistent_walk(ops, cfg):
    """Each tile configuration forced in turn: ragged edges, every epilogue, and a problem with more tiles than
    resident blocks so the persistent walk (next-tile prefetch, private epilogue scratch) is exercised."""
    from hri_emo_amd import _lib
    L = _lib.lib()
    L.hriemo_gemm_force_config(cfg)
    try:
        # (the 6- / 7-deep rings of configurations 6-8 need K > 320: shorter problems fall back to configuration 0 by design)
        for (M, N, K) in [(200, 136, 96), (300, 264, 160), (8192, 4096, 128), (390, 776, 512), (70, 2056, 1032)]:
            A, W, b = ints((M, K), seed=11), ints((N, K), seed=12), ints((N,), seed=13)
            ref = A @ W.t() + b
            y = ops.linear_fwd(A.cuda().bfloat16(), W.cuda().bfloat16(), b.cuda(), relu=True)
            assert torch.equal(y.float().cpu(), ref.clamp(min=0).bfloat16().float()), (cfg, M, N, K, "nt relu")
            dY, W2, aux = ints((M, N), seed=14), ints((N, K), seed=15), ints((M, K), seed=16)
            ref2 = dY @ W2
            dx2 = ops.linear_dx(dY.cuda().bfloat16(), W2.cuda().bfloat16(), epi=2, aux=aux.cuda().bfloat16())
            assert torch.equal(dx2.float().cpu(), (ref2 * (aux > 0)).bfloat16().float()), (cfg, M, N, K, "nn mask")
            dx3 = ops.linear_dx(dY.cuda().bfloat16(), W2.cuda().bfloat16(), epi=3, aux=aux.cuda().bfloat16())
            assert torch.equal(dx3.float().cpu(), (ref2 + aux).bfloat16().float()), (cfg, M, N, K, "nn add")
            X = ints((M, K), -2, 3, seed=17)
            dYs = ints((M, N), -2, 3, seed=18)
            out = torch.full((N, K), 1.0, dtype=torch.float32, device="cuda")
            ops.linear_dw(dYs.cuda().bfloat16(), X.cuda().bfloat16(), out)
            assert torch.equal(out.cpu(), dYs.t() @ X), (cfg, M, N, K, "tn")
    finally:
        L.hriemo_gemm_force_config(-1)


def test_colsum_and_cast(ops):
    X = ints((1000, 264), seed=9)
    out = torch.empty(264, dtype=torch.float32, device="cuda")
    ops.colsum(X.cuda().bfloat16(), out)
    assert torch.equal(out.cpu(), X.sum(0))
    p = torch.nn.Parameter(torch.randn(77, 13).cuda())
    sh = ops.Shadows()
    assert torch.equal(sh.get(p), p.detach().bfloat16())
    with torch.no_grad():
        p.add_(1.0)
    assert torch.equal(sh.get(p), p.detach().bfloat16())


# ------------------------------------------------------------------------------------------- attention
def ref_attention(q, k, v, kpm, keep, inv_keep):
    """q [B,H,Lq,hd] etc. fp32 (autograd-capable); keep [B,H,Lq,Lk] bool or None"""
    hd = q.shape[-1]
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    pd = p if keep is None else p * keep * inv_keep
    return pd @ v, torch.logsumexp(s, dim=-1), pd


ATTN_CASES = [  # B, H, Lq, Lk, hd, masked, p
    (2, 2, 6, 6, 16, False, 0.0),
    (2, 8, 6, 20, 96, True, 0.0),
    (2, 8, 48, 20, 96, True, 0.0),
    (2, 8, 20, 48, 96, True, 0.1),
    (1, 4, 130, 70, 64, True, 0.0),
    (2, 8, 400, 128, 96, True, 0.1),
    (2, 8, 128, 400, 96, False, 0.0),
    (3, 8, 32, 16, 16, True, 0.1),
    (1, 2, 200, 200, 128, False, 0.0),
    (1, 4, 50, 1000, 32, True, 0.0),
    (64, 8, 128, 70, 96, True, 0.1),      # B*H = 512 at L <= 128: the backward picks its 128-row tiles (one round of blocks)
    (2, 8, 6, 128, 96, True, 0.1),        # decoder cross-attention: N_e = 6 queries over the fused memory (single-pass backward)
    (2, 4, 100, 128, 128, True, 0.1),     # head_dim 128 (cfg 5), all 128 keys in one block
    (3, 4, 70, 40, 64, True, 0.1),        # L_k <= 64: one 16-key sub-tile per wave
    (2, 2, 33, 17, 32, False, 0.2),
    (2, 8, 400, 129, 96, True, 0.1),      # one key past the single-pass limit: two-kernel path with the bit-word mask
    (2, 8, 128, 400, 96, True, 0.1),      # query-resident single pass (L_q <= 128 < L_k): t2a at cfg 2, ragged keys, dropout
    (3, 4, 100, 300, 128, True, 0.1),     # ... head_dim 128, query tail (100 of 128 rows), key tail (300 = 4 tiles + 44)
    (2, 4, 17, 129, 64, False, 0.2),      # ... smallest shape that takes it
    (32, 8, 128, 400, 96, False, 0.1),    # ... every CU busy (256 blocks)
]


@pytest.mark.parametrize("B,H,Lq,Lk,hd,masked,p", ATTN_CASES)
def test_attention_fwd_bwd(ops, B, H, Lq, Lk, hd, masked, p):
    g = torch.Generator().manual_seed(100 + Lq + Lk)
    d = H * hd
    # projections laid out like the real buffers: q in [B*Lq, d], kv packed [B*Lk, 2d]
    qb = (torch.randn(B * Lq, d, generator=g) * 1.5).bfloat16()
    kvb = torch.randn(B * Lk, 2 * d, generator=g).bfloat16()
    dob = torch.randn(B * Lq, d, generator=g).bfloat16()
    kpm = None
    if masked:
        lens = torch.randint(max(1, Lk // 2), Lk + 1, (B,), generator=g)
        kpm = torch.arange(Lk)[None, :] >= lens[:, None]
    seed, site, boff = 1234567890123, 40, 5
    keep = None
    if p > 0:
        keep = torch.from_numpy(hashrng.attn_mask(seed, site, B, H, Lq, Lk, p, boff)).float()
    q = qb.float().view(B, Lq, H, hd).transpose(1, 2).detach().requires_grad_(True)
    k = kvb[:, :d].float().contiguous().view(B, Lk, H, hd).transpose(1, 2).detach().requires_grad_(True)
    v = kvb[:, d:].float().contiguous().view(B, Lk, H, hd).transpose(1, 2).detach().requires_grad_(True)
    o_ref, lse_ref, pd_ref = ref_attention(q, k, v, kpm, keep, hashrng.inv_keep(p))
    o_ref2 = o_ref.transpose(1, 2).reshape(B * Lq, d)
    o_ref2.backward(dob.float())

    qd, kvd, dod = qb.cuda(), kvb.cuda(), dob.cuda()
    kpm_d = kpm.cuda().view(torch.uint8) if kpm is not None else None
    o, lse, mbits = ops.attn_fwd(qd, kvd[:, :d], kvd[:, d:], B, H, Lq, Lk, hd, kpm_d, p, seed, site, boff, want_bits=True)
    tol = 2e-2     # bf16 P and bf16 O: ~2^-8 relative on O(1) values
    assert (o.float().cpu() - o_ref2.detach()).abs().max() <= tol * max(1.0, o_ref2.abs().max().item())
    assert (lse.cpu() - lse_ref.detach()).abs().max() <= 2e-3 * max(1.0, lse_ref.abs().max().item())
    if p > 0:
        # the keep-mask bit words the forward leaves for the backward: bit 16*g + 4*n + r of word (b, h, q, tile)
        # <-> key 64*tile + 16*n + 4*g + r, equal to the host replica of the hash
        nkt = (Lk + 63) // 64
        w = mbits.view(B, H, Lq, nkt).cpu().numpy().astype(np.uint64)
        key = np.arange(Lk)
        bitpos = ((key % 16) // 4) * 16 + ((key % 64) // 16) * 4 + key % 4
        got_keep = ((w[..., key // 64] >> bitpos.astype(np.uint64)) & np.uint64(1)).astype(bool)
        assert np.array_equal(got_keep, keep.numpy().astype(bool))
    else:
        assert mbits is None

    probs = ops.attn_probs(qd, kvd[:, :d], B, H, Lq, Lk, hd, kpm_d, lse, p, seed, site, boff)
    assert (probs.cpu() - pd_ref.detach().mean(1)).abs().max() <= 5e-3
    if kpm is not None:
        assert float(probs.cpu()[kpm[:, None, :].expand(B, Lq, Lk)].abs().max()) == 0.0

    dq = torch.empty_like(qd)
    dkv = torch.empty_like(kvd)
    ops.attn_bwd(qd, kvd[:, :d], kvd[:, d:], o, dod, dq, dkv[:, :d], dkv[:, d:], lse, B, H, Lq, Lk, hd, kpm_d, p, seed,
                 site, boff, mask_bits=mbits)
    dq_ref = q.grad.transpose(1, 2).reshape(B * Lq, d)
    dk_ref = k.grad.transpose(1, 2).reshape(B * Lk, d)
    dv_ref = v.grad.transpose(1, 2).reshape(B * Lk, d)
    for name, got, ref in (("dq", dq, dq_ref), ("dk", dkv[:, :d], dk_ref), ("dv", dkv[:, d:], dv_ref)):
        err = (got.float().cpu() - ref).abs().max().item()
        assert err <= 3e-2 * max(1.0, ref.abs().max().item()), (name, err, ref.abs().max().item())
    if p > 0:       # mask from the bit words == mask replayed from the hash: same kernels, bit-identical gradients
        dq_h, dkv_h = torch.empty_like(qd), torch.empty_like(kvd)
        ops.attn_bwd(qd, kvd[:, :d], kvd[:, d:], o, dod, dq_h, dkv_h[:, :d], dkv_h[:, d:], lse, B, H, Lq, Lk, hd, kpm_d, p, seed,
                     site, boff, mask_bits=None)
        assert torch.equal(dq_h, dq) and torch.equal(dkv_h, dkv)

    # by-product: per-block column sums of dQ and dK|dV (in-projection bias gradient) through the C-ABI
    from hri_emo_amd import _lib
    L_ = _lib.lib()
    rq, rk = L_.hriemo_attn_bwd_dq_colsum_rows(B, H, Lq, Lk, hd), L_.hriemo_attn_bwd_kv_colsum_rows(B, H, Lq, Lk, hd)
    pq = torch.full((rq, d), float("nan"), device="cuda")
    pkv = torch.full((rk, 2 * d), float("nan"), device="cuda")
    dq2, dkv2 = torch.empty_like(qd), torch.empty_like(kvd)
    delta = torch.empty_like(lse)
    k_, v_ = kvd[:, :d], kvd[:, d:]
    _lib.call("hriemo_attn_bwd", qd.data_ptr(), qd.stride(0), k_.data_ptr(), k_.stride(0), v_.data_ptr(), v_.stride(0),
              o.data_ptr(), o.stride(0), dod.data_ptr(), dod.stride(0), dq2.data_ptr(), dq2.stride(0),
              dkv2[:, :d].data_ptr(), dkv2.stride(0), dkv2[:, d:].data_ptr(), dkv2.stride(0),
              kpm_d.data_ptr() if kpm_d is not None else None, lse.data_ptr(), delta.data_ptr(), B, H, Lq, Lk, hd, float(p),
              seed, ops.seed_word(qd.device).data_ptr(), site, boff, pq.data_ptr(), pkv.data_ptr(),
              mbits.data_ptr() if mbits is not None else None, torch.cuda.current_stream().cuda_stream)
    assert torch.equal(dq2, dq) and torch.equal(dkv2, dkv)
    # the partials are sums of the fp32 values BEFORE their bf16 rounding: compared with the column sums of the fp32 reference
    # gradients (tolerance: the bf16 P / dS operands of the kernels, summed over B*L rows), and they must be at least as close to
    # it as the column sums of the stored (rounded) tiles are
    for name, part, full, ref in (("dq", pq, dq, dq_ref.sum(0)), ("dkv", pkv, dkv, torch.cat([dk_ref, dv_ref], 1).sum(0))):
        got = part.sum(0).cpu()
        stored = full.float().sum(0).cpu()
        assert not torch.isnan(part).any(), name       # every partial row is written
        scale = max(1.0, ref.abs().max().item())
        assert (got - ref).abs().max() <= 2e-2 * scale, (name, (got - ref).abs().max(), scale)
        assert (got - stored).abs().max() <= 1e-2 * scale, (name, "vs stored tiles", (got - stored).abs().max())
        assert (got - ref).norm() <= 1.05 * (stored - ref).norm() + 1e-6 * scale, (name, (got - ref).norm(), (stored - ref).norm())


def test_attention_all_pad_row_is_nan(ops):
    B, H, L, hd = 2, 2, 8, 16
    d = H * hd
    q = torch.randn(B * L, d).bfloat16().cuda()
    kv = torch.randn(B * L, 2 * d).bfloat16().cuda()
    kpm = torch.zeros(B, L, dtype=torch.bool)
    kpm[1, :] = True
    o, lse = ops.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, L, L, hd, kpm.cuda().view(torch.uint8), 0.0, 0, 0, 0)
    o = o.float().cpu().view(B, L, d)
    assert torch.isnan(o[1]).all() and not torch.isnan(o[0]).any()
    # the exported map agrees with O and with the reference's need_weights=True path: softmax over an all -inf row is NaN
    probs = ops.attn_probs(q, kv[:, :d], B, H, L, L, hd, kpm.cuda().view(torch.uint8), lse, 0.0, 0, 0, 0).cpu()
    assert torch.isnan(probs[1]).all() and torch.isfinite(probs[0]).all()
    assert (probs[0].sum(-1) - 1).abs().max() <= 1e-3


# ------------------------------------------------------------------------------------------- add + LN
@pytest.mark.parametrize("M,d,p,resid", [(37, 128, 0.0, True), (300, 768, 0.1, True), (64, 768, 0.3, False),
                                         (10, 1024, 0.1, True), (9, 2048, 0.0, True)])
def test_add_ln_fwd_bwd(ops, M, d, p, resid):
    g = torch.Generator().manual_seed(7 + M)
    G = torch.randn(M, d, generator=g).bfloat16()
    X = torch.randn(M, d, generator=g).bfloat16() if resid else None
    gamma = (1 + 0.1 * torch.randn(d, generator=g))
    beta = 0.1 * torch.randn(d, generator=g)
    dY = torch.randn(M, d, generator=g).bfloat16()
    seed, site, roff = 987654321, 12, 1000
    keep = torch.from_numpy(hashrng.rows_mask(seed, site, M, d, p, roff)).float() if p > 0 else torch.ones(M, d)
    ik = hashrng.inv_keep(p)
    Gf = G.float().requires_grad_(True)
    Xf = X.float().requires_grad_(True) if resid else None
    gam, bet = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    s = Gf * keep * ik + (Xf if resid else 0)
    y_ref = torch.nn.functional.layer_norm(s, (d,), gam, bet, 1e-5)
    y_ref.backward(dY.float())

    y, y32, mean, rstd = ops.add_ln_fwd(G.cuda(), X.cuda() if resid else None, gamma.cuda(), beta.cuda(), p, seed, site, roff,
                                        want32=True)
    assert (y.float().cpu() - y_ref.detach()).abs().max() <= 2e-2      # bf16 output of O(1..4) values
    assert (y32.cpu() - y_ref.detach()).abs().max() <= 1e-4            # fp32 twin of the same LayerNorm output
    assert (mean.cpu() - s.detach().mean(-1)).abs().max() <= 1e-5
    if resid:                                                          # fp32 residual twin as the input
        y2, y2_32, _, _ = ops.add_ln_fwd(G.cuda(), None, gamma.cuda(), beta.cuda(), p, seed, site, roff,
                                         x32=X.float().cuda(), want32=True)
        # (this call takes the quad-mapped kernel, the bf16-residual one above the chunk-mapped: other summation order)
        assert (y2_32 - y32).abs().max() <= 2e-6 * max(1.0, y32.abs().max().item()) and (y2.float() - y.float()).abs().max() <= 4e-2
    dx, dg, dgam, dbet, dbias = ops.add_ln_bwd(dY.cuda(), G.cuda(), X.cuda() if resid else None, gamma.cuda(), mean, rstd,
                                               p, seed, site, roff)
    assert (dg.float().cpu() - Gf.grad).abs().max() <= 2e-2 * max(1.0, Gf.grad.abs().max().item())
    if resid:
        assert (dx.float().cpu() - Xf.grad).abs().max() <= 2e-2 * max(1.0, Xf.grad.abs().max().item())
    assert (dgam.cpu() - gam.grad).abs().max() <= 2e-3 * max(1.0, gam.grad.abs().max().item())
    assert (dbet.cpu() - bet.grad).abs().max() <= 2e-3 * max(1.0, bet.grad.abs().max().item())
    assert (dbias.cpu() - Gf.grad.sum(0)).abs().max() <= 2e-2 * max(1.0, Gf.grad.sum(0).abs().max().item())


@pytest.mark.parametrize("M,d,p", [(37, 128, 0.1), (1000, 768, 0.1), (64, 768, 0.0), (4100, 768, 0.1), (333, 1024, 0.2), (50, 256, 0.1),
                                   (21, 512, 0.1), (5, 704, 0.1)])
def test_add_ln_quad_mapped_kernels_equal_the_chunk_mapped_ones(ops, M, d, p):
    """Round 4: LayerNorm(x + dropout(g)) forward / backward on the quad mapping (lane l owns 4-element quads l, l + 64, ...;
    software-pipelined rows, persistent grid; d <= 1024, fp32 twin in and out) against the chunk-mapped kernels on the same
    operands through the C-ABI switch hriemo_rowops_force_variant: the SAME dropout masks (outputs equal to fp32 rounding of
    the two wave sums, dropped elements exactly zero in both), the same column sums, and both against fp32 torch math.
    Shapes: d = 768 / 1024 / 256 / 512 (the widths the quad mapping is built for), d = 128 and 704 (not multiples of 256: both
    settings run the chunk-mapped kernel and must agree bit for bit), M below and above one grid round."""
    from hri_emo_amd import _lib
    g = torch.Generator().manual_seed(11 + M + d)
    G = torch.randn(M, d, generator=g).bfloat16().cuda()
    X32 = torch.randn(M, d, generator=g).cuda()
    gamma = (1 + 0.1 * torch.randn(d, generator=g)).cuda()
    beta = (0.1 * torch.randn(d, generator=g)).cuda()
    dY = torch.randn(M, d, generator=g).bfloat16().cuda()
    seed, site, roff = 24681357, 8, 4242
    res = {}
    try:
        for variant in (1, 0):
            _lib.call("hriemo_rowops_force_variant", variant)
            y, y32, mean, rstd = ops.add_ln_fwd(G, None, gamma, beta, p, seed, site, roff, x32=X32, want32=True)
            dx, dg, dgam, dbet, dbias = ops.add_ln_bwd(dY, G, None, gamma, mean, rstd, p, seed, site, roff, x32=X32)
            torch.cuda.synchronize()
            res[variant] = [t.float().cpu() for t in (y, y32, mean, rstd, dx, dg, dgam, dbet, dbias)]
    finally:
        _lib.call("hriemo_rowops_force_variant", 1)
    keep = torch.from_numpy(hashrng.rows_mask(seed, site, M, d, p, roff)).float() if p > 0 else torch.ones(M, d)
    Gf = G.float().cpu().requires_grad_(True)
    Xf = X32.cpu().clone().requires_grad_(True)
    gam, bet = gamma.cpu().clone().requires_grad_(True), beta.cpu().clone().requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(Gf * keep * hashrng.inv_keep(p) + Xf, (d,), gam, bet, 1e-5)
    y_ref.backward(dY.float().cpu())
    names = ("y", "y32", "mean", "rstd", "dx", "dg", "dgamma", "dbeta", "dbias")
    tol16, tol32 = 2e-2, 1e-4
    for v in (0, 1):
        y, y32, mean, rstd, dx, dg, dgam, dbet, dbias = res[v]
        assert (y32 - y_ref.detach()).abs().max() <= tol32, (v, "y32")
        assert (y - y_ref.detach()).abs().max() <= tol16 * max(1.0, y_ref.abs().max().item()), (v, "y")
        assert (dx - Xf.grad).abs().max() <= tol16 * max(1.0, Xf.grad.abs().max().item()), (v, "dx")
        assert (dg - Gf.grad).abs().max() <= tol16 * max(1.0, Gf.grad.abs().max().item()), (v, "dg")
        assert ((dg == 0) | (keep > 0)).all(), (v, "a dropped element carries gradient")
        assert (dgam - gam.grad).abs().max() <= 2e-3 * max(1.0, gam.grad.abs().max().item()), (v, "dgamma")
        assert (dbet - bet.grad).abs().max() <= 2e-3 * max(1.0, bet.grad.abs().max().item()), (v, "dbeta")
        assert (dbias - Gf.grad.sum(0)).abs().max() <= 2e-2 * max(1.0, Gf.grad.sum(0).abs().max().item()), (v, "dbias")
    for n, a, b in zip(names, res[0], res[1]):
        lim = 1e-5 if n in ("y32", "mean", "rstd") else (2e-2 if n in ("y", "dx", "dg") else 1e-4)      # bf16 outputs: one rounding apart at most
        assert (a - b).abs().max() <= lim * max(1.0, b.abs().max().item()), (n, float((a - b).abs().max()))
    frac_equal = float((res[0][0] == res[1][0]).float().mean())
    assert frac_equal > 0.99, frac_equal            # bf16 y: the mappings agree bit for bit almost everywhere
    if d % 256:
        for n, a, b in zip(names, res[0], res[1]):
            assert torch.equal(a, b), n


def test_rowdot_expand(ops):
    B, Ne, d = 5, 6, 768
    g = torch.Generator().manual_seed(3)
    z = torch.randn(B, Ne, d, generator=g).bfloat16().cuda().requires_grad_(True)
    w = torch.nn.Parameter(torch.randn(1, d, generator=g).cuda())
    b = torch.nn.Parameter(torch.randn(1, generator=g).cuda())
    out = ops.RowDotFn.apply(z, None, w, b)
    ref = (z.float() @ w.t()).squeeze(-1) + b
    assert (out - ref).abs().max() <= 1e-3 * max(1.0, ref.abs().max().item())
    dl = torch.randn(B, Ne, generator=g).cuda()
    out.backward(dl)
    assert (z.grad.float() - dl[..., None] * w).abs().max() <= 2e-2
    assert (w.grad - (dl.view(-1, 1) * z.detach().float().view(-1, d)).sum(0, keepdim=True)).abs().max() <= 1e-3 * d ** 0.5
    assert (b.grad - dl.sum()).abs().max() <= 1e-4
    q = torch.nn.Parameter(torch.randn(Ne, d, generator=g).cuda())
    e = ops.ExpandFn.apply(q, B)
    assert torch.equal(e, q.detach().bfloat16()[None].expand(B, Ne, d))
    go = torch.randn(B, Ne, d, generator=g).bfloat16().cuda()
    e.backward(go)
    assert (q.grad - go.float().sum(0)).abs().max() <= 1e-4


# ------------------------------------------------------------------------------------------- Linear + LayerNorm in one kernel
@pytest.mark.parametrize("M,d,K,p,twin,mapped", [(1100, 768, 768, 0.1, True, False), (2048, 768, 3072, 0.0, True, False),
                                                 (130, 256, 128, 0.1, False, False), (64, 512, 96, 0.3, True, False),
                                                 (333, 768, 768, 0.1, False, True), (4096, 768, 768, 0.1, True, True)])
def test_gemm_ln_fused_equals_gemm_then_add_ln(ops, M, d, K, p, twin, mapped):
    """hriemo_gemm_ln_fwd (csrc/gemm_ln.hip: Linear + bias + dropout + residual + LayerNorm on full-row tiles) against the two
    launches it replaces, hriemo_gemm_bf16 then hriemo_add_ln_fwd(_rows), on the same operands: g (what the backward reads)
    bit-identical, the same dropout mask (a dropped element changes y by O(1)), y / y32 / mean / rstd equal to the rounding of
    the row sums (the statistics are reduced in another order: 8 waves x 4 lane groups instead of one wave).  Ragged M (not a
    multiple of the 64-row tile), K not a multiple of the 32-deep stage, bf16 or fp32 residual, packed-row dropout keys."""
    g = torch.Generator().manual_seed(M + d + K)
    A = (0.5 * torch.randn(M, K, generator=g)).bfloat16().cuda()
    W = (torch.randn(d, K, generator=g) / math.sqrt(K)).bfloat16().cuda()
    b = (0.1 * torch.randn(d, generator=g)).cuda()
    X32 = torch.randn(M, d, generator=g).cuda()
    X16 = X32.bfloat16()
    gamma = (1 + 0.1 * torch.randn(d, generator=g)).cuda()
    beta = (0.1 * torch.randn(d, generator=g)).cuda()
    rows = (torch.randperm(3 * M, generator=g)[:M].sort().values.to(torch.int64).cuda()) if mapped else None
    seed, site, roff = 13572468, 5, 640
    x32 = X32 if twin else None
    g_ref = ops.linear_fwd(A, W, b)
    y_ref, y32_ref, mean_ref, rstd_ref = ops.add_ln_fwd(g_ref, X16, gamma, beta, p, seed, site, roff, x32=x32, want32=True, rows=rows)
    g_f, y_f, y32_f, mean_f, rstd_f = ops.proj_add_ln_fwd(A, W, b, X16, x32, gamma, beta, p, seed, site, roff, True, rows)
    torch.cuda.synchronize()
    assert torch.equal(g_f, g_ref)
    scale = max(1.0, float(y32_ref.abs().max()))
    assert float((y32_f - y32_ref).abs().max()) <= 4e-6 * scale, float((y32_f - y32_ref).abs().max())
    assert float((y_f.float() - y_ref.float()).abs().max()) <= 2 ** -6 * scale            # one bf16 rounding of O(1..4) values
    assert float((y_f.float() != y_ref.float()).float().mean()) <= 1e-3                   # and that only where a value sits on a tie
    assert float((mean_f - mean_ref).abs().max()) <= 1e-6 and float((rstd_f / rstd_ref - 1).abs().max()) <= 1e-5
    # against fp32 torch math with the host replica of the mask
    if rows is None and p > 0:
        keep = torch.from_numpy(hashrng.rows_mask((seed + int(ops.seed_word(torch.device("cuda", 0)).item())) & ((1 << 64) - 1),
                                                  site, M, d, p, roff)).cuda()
        s = g_ref.float() * keep * hashrng.inv_keep(p) + (X32 if twin else X16.float())
        y_t = torch.nn.functional.layer_norm(s, (d,), gamma, beta, 1e-5)
        assert float((y32_f - y_t).abs().max()) <= 1e-4 * scale
    # no fp32 twin asked for, no g kept: both optional outputs may be absent
    _, y_n, y32_n, _, _ = ops.proj_add_ln_fwd(A, W, b, X16, x32, gamma, beta, p, seed, site, roff, False, rows)
    assert y32_n is None and torch.equal(y_n, y_f)


# ------------------------------------------------------------------------------------------- loader / consumer GEMM, both walks
@pytest.mark.parametrize("flags", [9, 1])
def test_gemm_loader_consumer_kernel_static_walk_and_work_queue(ops, flags):
    """Configuration 9 (gemm_ws_kernel: 8 MFMA waves + 4 waves that only stage operands) on problems with several rounds of
    tiles, with the static walk (hriemo_gemm_debug_flags bit 3 set: the N = 1 default) and with the per-XCD work queue (bit 3
    clear: what dp.py selects while collectives run beside backward; ids handed from consumer wave 0 to the other eleven waves
    through its idle epilogue scratch): NT with bias / ReLU, NN with the residual epilogue, the fp32 weight-gradient layout with
    split-K, ragged edges, units with two K-steps (the queue's shortest hand-over).  Exact on integers -- a tile that is skipped,
    computed twice or started from a stale id shows at once."""
    from hri_emo_amd import _lib
    L = _lib.lib()
    prev = L.hriemo_gemm_debug_flags(flags)
    L.hriemo_gemm_force_config(9)
    try:
        for (M, N, K) in [(25600, 768, 768), (70000, 768, 128), (66000, 264, 192), (5000, 776, 1024), (40000, 1536, 256)]:
            A, W, b = ints((M, K), seed=1), ints((N, K), seed=2), ints((N,), seed=3)
            Ad, Wd = A.cuda().bfloat16(), W.cuda().bfloat16()
            ref = A @ W.t() + b
            assert torch.equal(ops.linear_fwd(Ad, Wd, b.cuda()).float().cpu(), ref.bfloat16().float()), (flags, "NT", M, N, K)
            assert torch.equal(ops.linear_fwd(Ad, Wd, b.cuda(), relu=True).float().cpu(), ref.clamp(min=0).bfloat16().float()), (flags, "NT relu")
            dY, W2, R = ints((M, N), seed=5), ints((N, K), seed=6), ints((M, K), seed=7)
            dx = ops.linear_dx(dY.cuda().bfloat16(), W2.cuda().bfloat16(), epi=3, aux=R.cuda().bfloat16())
            assert torch.equal(dx.float().cpu(), (dY @ W2 + R).bfloat16().float()), (flags, "NN + aux", M, N, K)
        for (Nout, Kout, Mred) in [(768, 768, 25600), (3072, 768, 8192), (776, 264, 12000)]:
            dY, X = ints((Mred, Nout), -2, 3, seed=8), ints((Mred, Kout), -2, 3, seed=9)
            out = torch.full((Nout, Kout), 3.0, device="cuda")
            ops.linear_dw(dY.cuda().bfloat16(), X.cuda().bfloat16(), out, True)
            assert torch.equal(out.cpu(), dY.t() @ X + 3.0), (flags, "TN", Nout, Kout, Mred)
    finally:
        L.hriemo_gemm_force_config(-1)
        L.hriemo_gemm_debug_flags(prev)
