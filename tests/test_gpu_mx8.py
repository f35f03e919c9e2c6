"""GPU suite: the MX-fp8 operand path of BASELINE.json configs[4] ("fp8 MFMA path": d=1024, 4+2 layers, N_e=7).

 * quantiser vs the host emulation (tests/mx8_emul.py): element bytes and scale bytes identical;
 * scaled-MFMA GEMM with integer operands that the format holds exactly: results EQUAL the integer matmul for every
   tile configuration, ragged edges, odd step counts, every epilogue -- a transposed or permuted fragment, a scale
   byte applied to the wrong block, or a wrong k order cannot hide;
 * GEMM on random data vs the emulated operands multiplied in fp64;
 * the module at cfg-5 dimensions against the fp32 oracle with the tolerance stated below and the oracle-with-
   emulated-fp8-operands as the yardstick (same method as the bf16 autocast yardstick)."""
import math

import pytest
import torch

from mx8_emul import mx8_dequantize, mx8_quantize, mx8_roundtrip
from oracle import hri_emo_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import hri_emo_amd  # noqa: F401
    from hri_emo_amd import _ops
    return _ops


def ints(shape, lo=-8, hi=9, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).float()


@pytest.mark.parametrize("M,K,src", [(64, 128, "bf16"), (300, 1024, "bf16"), (37, 4096, "bf16"), (1024, 768, "f32"), (8, 32, "f32")])
def test_quantiser_matches_host_emulation_bitwise(ops, M, K, src):
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g) * 2)      # rows of very different magnitude
    x[0, :32] = 0.0                                                                          # an all-zero block
    x[min(1, M - 1), 5] = 448.0 * 4                                                          # a block whose maximum sits on a scale boundary
    if src == "bf16":
        x = x.bfloat16()
    q, sc = ops.quant_mx8(x.cuda())
    q_ref, s_ref = mx8_quantize(x.float())
    assert torch.equal(q.cpu(), q_ref), (q.cpu() != q_ref).sum()
    assert sc.shape[0] == K // 32 and sc.shape[1] % 256 == 0 and sc.shape[1] >= M
    assert torch.equal(sc[:, :M].t().cpu(), s_ref)
    back = mx8_dequantize(q.cpu(), sc[:, :M].t().cpu())
    rel = ((back - x.float()).abs() / x.float().abs().amax(-1, keepdim=True).clamp_min(1e-30)).max().item()
    assert rel <= 2 ** -4, rel                            # e4m3: 3 mantissa bits, and the block maximum never saturates


@pytest.mark.parametrize("M,d,p", [(300, 768, 0.1), (1030, 1024, 0.0), (64, 128, 0.1), (17, 2048, 0.0)])
def test_layernorm_fused_mx8_copy_equals_separate_quantiser(ops, M, d, p):
    """hriemo_add_ln_fwd_mx8: the MX-fp8 copy written by the LayerNorm kernel is bit-identical to hriemo_quant_mx8 of its
    bf16 output, and the bf16 / fp32 outputs are unchanged by asking for it."""
    g = torch.Generator().manual_seed(M + d)
    G, X = torch.randn(M, d, generator=g).bfloat16().cuda(), (torch.randn(M, d, generator=g) * 2).bfloat16().cuda()
    gamma, beta = (1 + 0.1 * torch.randn(d, generator=g)).cuda(), (0.1 * torch.randn(d, generator=g)).cuda()
    y0, y032, mean0, rstd0 = ops.add_ln_fwd(G, X, gamma, beta, p, 1234, 7, 100, want32=True)
    y1, y132, mean1, rstd1, (yq, ys) = ops.add_ln_fwd(G, X, gamma, beta, p, 1234, 7, 100, want32=True, want_mx=True)
    assert torch.equal(y0, y1) and torch.equal(y032, y132) and torch.equal(mean0, mean1) and torch.equal(rstd0, rstd1)
    q2, s2 = ops.quant_mx8(y1)
    assert torch.equal(yq, q2) and torch.equal(ys[:, :M], s2[:, :M])
    q_ref, s_ref = mx8_quantize(y1.float().cpu())
    assert torch.equal(yq.cpu(), q_ref) and torch.equal(ys[:, :M].t().cpu(), s_ref)


MX_SHAPES = [(256, 128, 128), (200, 136, 256), (128, 384, 384), (1024, 768, 768), (64, 256, 3072), (37 * 8, 8, 128),
             (130, 2304, 768), (8192, 1024, 1024), (300, 264, 640)]


@pytest.mark.parametrize("cfg", [-1, 0, 1, 2])
@pytest.mark.parametrize("M,N,K", MX_SHAPES)
def test_gemm_mx8_exact_on_integer_operands(ops, M, N, K, cfg):
    """integers in [-8, 8] survive the quantiser exactly (x * 2^-e is an integer with <= 4 significant bits), so the scaled-MFMA
    result must equal the integer matmul: bf16 output (rounded once), ReLU, and fp32 output."""
    from hri_emo_amd import _lib
    L = _lib.lib()
    L.hriemo_gemm_mx8_force_config(cfg)
    try:
        A, W, b = ints((M, K), seed=1), ints((N, K), -6, 7, seed=2), ints((N,), -3, 4, seed=3)
        # blocks with different maxima -> different scale bytes per (row, k-block)
        A[:, :32] = A[:, :32].clamp(-2, 2)
        W[:, 32:64] = W[:, 32:64].clamp(-1, 1)
        ref = A @ W.t() + b
        aq, as_ = ops.quant_mx8(A.cuda().bfloat16())
        wq, ws = ops.quant_mx8(W.cuda())                      # fp32 source, like the weight masters
        assert torch.equal(mx8_dequantize(aq.cpu(), as_[:, :M].t().cpu()), A)
        y = ops.linear_fwd_mx8(aq, as_, wq, ws, b.cuda())
        assert torch.equal(y.float().cpu(), ref.bfloat16().float()), (cfg, (y.float().cpu() - ref.bfloat16().float()).abs().max())
        yr = ops.linear_fwd_mx8(aq, as_, wq, ws, b.cuda(), relu=True)
        assert torch.equal(yr.float().cpu(), ref.clamp(min=0).bfloat16().float())
        yf = ops.linear_fwd_mx8(aq, as_, wq, ws, b.cuda(), out_f32=True)
        assert torch.equal(yf.cpu(), ref)
    finally:
        L.hriemo_gemm_mx8_force_config(-1)


def test_gemm_mx8_loader_consumer_kernel_work_queue(ops):
    """gemm_mx8_ws_kernel on several rounds of tiles with the per-XCD work queue (hriemo_gemm_debug_flags bit 3 clear) and with
    the static walk: exact on integers (see test_gemm_mx8_exact_on_integer_operands)"""
    from hri_emo_amd import _lib
    L = _lib.lib()
    L.hriemo_gemm_mx8_force_config(2)
    for flags in (1, 9):
        prev = L.hriemo_gemm_debug_flags(flags)
        try:
            for (M, N, K) in [(25600, 1024, 256), (40000, 768, 512), (9000, 2056, 384)]:
                A, W, b = ints((M, K), seed=1), ints((N, K), -6, 7, seed=2), ints((N,), -3, 4, seed=3)
                ref = A @ W.t() + b
                aq, as_ = ops.quant_mx8(A.cuda().bfloat16())
                wq, ws = ops.quant_mx8(W.cuda())
                y = ops.linear_fwd_mx8(aq, as_, wq, ws, b.cuda())
                assert torch.equal(y.float().cpu(), ref.bfloat16().float()), (flags, M, N, K)
        finally:
            L.hriemo_gemm_debug_flags(prev)
    L.hriemo_gemm_mx8_force_config(-1)


@pytest.mark.parametrize("cfg", [-1, 0, 1, 2])
@pytest.mark.parametrize("M,N,K,relu", [(1000, 512, 256, True), (12800, 4096, 1024, True), (300, 128, 640, False), (8192, 1024, 1024, False)])
def test_gemm_mx8_epilogue_leaves_the_quantised_output(ops, M, N, K, relu, cfg):
    """hriemo_gemm_mx8_q (round 4): the epilogue that stores the bf16 tile also writes its MX-fp8 form -- bytes and E8M0 scales must
    be bit-identical to hriemo_quant_mx8 of the stored output (what FFN2 used to read from a separate pass), on full and edge
    tiles, with and without ReLU; the bf16 output itself must equal the plain launch's"""
    from hri_emo_amd import _lib
    L = _lib.lib()
    L.hriemo_gemm_mx8_force_config(cfg)
    try:
        g = torch.Generator().manual_seed(M + N)
        A = (torch.randn(M, K, generator=g) * torch.exp(0.5 * torch.randn(M, 1, generator=g))).bfloat16().cuda()
        W = (torch.randn(N, K, generator=g) / math.sqrt(K)).cuda()
        b = torch.randn(N, generator=g).cuda()
        aq, as_ = ops.quant_mx8(A)
        wq, ws = ops.quant_mx8(W)
        y0 = ops.linear_fwd_mx8(aq, as_, wq, ws, b, relu=relu)
        y1 = ops.linear_fwd_mx8(aq, as_, wq, ws, b, relu=relu, want_q=True)
        q1, s1 = ops.mx_of(y1)
        assert torch.equal(y0, y1)
        q2, s2 = ops.quant_mx8(y0)
        assert torch.equal(q1, q2), float((q1 != q2).float().mean())
        assert torch.equal(s1[:, :M], s2[:, :M])
    finally:
        L.hriemo_gemm_mx8_force_config(-1)


@pytest.mark.parametrize("B,H,Lq,Lk,hd,p", [(8, 8, 400, 128, 128, 0.1), (16, 8, 128, 400, 128, 0.0), (10, 8, 130, 70, 64, 0.1), (40, 4, 50, 33, 32, 0.1),
                                            (16, 8, 100, 100, 96, 0.1)])
def test_attention_forward_leaves_the_quantised_output(ops, monkeypatch, B, H, Lq, Lk, hd, p):
    """hriemo_attn_fwd_q (round 4, fp8 GEMM mode): the attention forward's epilogue also writes the MX-fp8 form of O -- bit-identical
    to hriemo_quant_mx8 of the O it stores (ragged keys, dropout, query tails, both tile widths); O and lse equal the plain launch's"""
    d = H * hd
    g = torch.Generator().manual_seed(B + Lq)
    q = torch.randn(B * Lq, d, generator=g).bfloat16().cuda()
    kv = torch.randn(B * Lk, 2 * d, generator=g).bfloat16().cuda()
    lk = torch.randint(1, Lk + 1, (B,), generator=g)
    kpm = (torch.arange(Lk)[None] >= lk[:, None]).cuda().view(torch.uint8)
    o0, lse0 = ops.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, Lq, Lk, hd, kpm, p, 77, 3, 0)
    assert ops.mx_of(o0) is None
    monkeypatch.setattr(ops, "GEMM_MODE", "mx_fp8")
    if d % 128 == 0 and B * Lq >= ops.MX_MIN_ROWS:
        o1, lse1 = ops.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, Lq, Lk, hd, kpm, p, 77, 3, 0)
        assert torch.equal(o0, o1) and torch.equal(lse0, lse1)
        oq, so = ops.mx_of(o1)
        q2, s2 = ops.quant_mx8(o0)
        M = B * Lq
        assert torch.equal(oq, q2), float((oq != q2).float().mean())
        assert torch.equal(so[:, :M], s2[:, :M])
    else:
        o1, _ = ops.attn_fwd(q, kv[:, :d], kv[:, d:], B, H, Lq, Lk, hd, kpm, p, 77, 3, 0)
        assert ops.mx_of(o1) is None and torch.equal(o0, o1)


def test_gemm_mx8_random_operands_vs_emulated_product(ops):
    g = torch.Generator().manual_seed(77)
    M, N, K = 1000, 520, 1024
    A = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).bfloat16()
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    aq, as_ = ops.quant_mx8(A.cuda())
    wq, ws = ops.quant_mx8(W.cuda())
    y = ops.linear_fwd_mx8(aq, as_, wq, ws, b.cuda(), out_f32=True).cpu()
    Ad, Wd = mx8_roundtrip(A.float()).double(), mx8_roundtrip(W).double()
    ref = (Ad @ Wd.t() + b.double()).float()
    assert (y - ref).abs().max() <= 1e-4 * max(1.0, ref.abs().max().item())      # same operands, fp32 accumulation order only
    true = A.float() @ W.t() + b
    rel = ((y - true).norm() / true.norm()).item()
    assert rel <= 6e-2, rel                                                      # what e4m3 operands cost on this product


# --------------------------------------------------------------------------------------------- module level
def _rand_batch(B, Ta, Tt, d, seed):
    g = torch.Generator().manual_seed(seed)
    h_a, h_t = torch.randn(B, Ta, d, generator=g), torch.randn(B, Tt, d, generator=g)
    la = torch.randint(max(1, Ta // 2), Ta + 1, (B,), generator=g)
    lt = torch.randint(max(1, Tt // 2), Tt + 1, (B,), generator=g)
    return h_a, h_t, torch.arange(Ta)[None] >= la[:, None], torch.arange(Tt)[None] >= lt[:, None]


def _err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / max(1.0, b.abs().max().item())).item()


@pytest.mark.parametrize("B,Ta,Tt,d,ne,lf,ld", [(2, 400, 128, 1024, 7, 4, 2),       # BASELINE configs[4] dimensions
                                                (3, 100, 40, 768, 6, 2, 2), (4, 32, 16, 128, 4, 2, 2)])
def test_fusion_mx_fp8_vs_oracle_and_emulated_yardstick(ops, monkeypatch, B, Ta, Tt, d, ne, lf, ld):
    """Stated tolerance of the fp8 GEMM mode: max|out - fp32 oracle| <= 6e-2 * max(1, max|ref|) on logits / beta / z, AND
    within 2.5x the yardstick: the oracle evaluated with the SAME quantiser on both operands of every projection / FFN GEMM
    (oracle.LINEAR_OPERAND_HOOK = mx8_roundtrip), whose own distance from the fp32 oracle is what e4m3 operands cost the
    reference path itself.  The HIP result must also stay within 1.5x that distance of the emulated reference."""
    import hri_emo_amd as H
    monkeypatch.setattr(ops, "MX_MIN_ROWS", 1)      # the product gates fp8 on >= 1024 rows (throughput-bound GEMMs); here every GEMM takes it
    torch.manual_seed(1234)
    kw = dict(d_model=d, num_emotions=ne, n_heads=8, dropout=0.1, num_layers_fusion=lf, num_layers_decoder=ld)
    ref = O.FusionWithEmotionDecoder(**kw).eval()
    m = H.FusionWithEmotionDecoder(**kw)
    m.load_state_dict(ref.state_dict())
    m.cuda().eval()
    h_a, h_t, m_a, m_t = _rand_batch(B, Ta, Tt, d, 5)
    with torch.no_grad():
        out32 = ref(h_a, h_t, m_a, m_t)
        O.LINEAR_OPERAND_HOOK = mx8_roundtrip
        try:
            out8 = ref(h_a, h_t, m_a, m_t)
        finally:
            O.LINEAR_OPERAND_HOOK = None
        H.set_gemm_mode("mx_fp8")
        try:
            got = m(h_a.cuda(), h_t.cuda(), m_a.cuda(), m_t.cuda())
        finally:
            H.set_gemm_mode("bf16")
        got16 = m(h_a.cuda(), h_t.cuda(), m_a.cuda(), m_t.cuda())
    for name, g8, g16, r32, r8 in zip(("logits", "beta", "z"), got, got16, out32, out8):
        yard = _err(r8, r32)
        e32, e8 = _err(g8, r32), _err(g8, r8)
        assert e32 <= 6e-2, (name, "vs fp32 oracle", e32, "yardstick (emulated fp8 oracle vs fp32 oracle)", yard)
        assert e32 <= max(1e-2, 2.5 * yard), (name, e32, yard)
        # the HIP path quantises bf16 activations, the emulated oracle fp32 ones: a fraction of the e4m3 roundings flip, so the two
        # fp8 realisations differ by about the quantisation noise itself (the yardstick), not by the bf16 mode's 2e-3
        assert e8 <= max(1.5e-2, 1.5 * yard), (name, "vs emulated-fp8 oracle", e8, "yardstick", yard, "bf16 mode", _err(g16, r32))
        assert not torch.equal(g8.float().cpu(), g16.float().cpu()) or d % 128 != 0, "fp8 mode must actually change the GEMMs"


def test_fusion_mx_fp8_trains(ops, monkeypatch):
    """forward on fp8 operands, backward on the bf16 GEMMs: gradients finite and close to the bf16 mode's (straight-through)."""
    import hri_emo_amd as H
    monkeypatch.setattr(ops, "MX_MIN_ROWS", 1)
    torch.manual_seed(7)
    m = H.FusionWithEmotionDecoder(d_model=256, num_emotions=4, n_heads=8, dropout=0.0).cuda().train()
    h_a, h_t, m_a, m_t = _rand_batch(4, 48, 24, 256, 9)
    y = (torch.rand(4, 4, generator=torch.Generator().manual_seed(1)) < 0.3).float().cuda()
    args = (h_a.cuda(), h_t.cuda(), m_a.cuda(), m_t.cuda())

    def grads():
        m.zero_grad()
        logits, beta, _ = m(*args)
        O.train_step_loss(logits, beta, y).backward()
        return {n: p.grad.detach().clone() for n, p in m.named_parameters()}

    g16 = grads()
    H.set_gemm_mode("mx_fp8")
    try:
        g8 = grads()
    finally:
        H.set_gemm_mode("bf16")
    rels = sorted(((g8[n] - g16[n]).norm() / g16[n].norm().clamp_min(1e-30)).item() for n in g16)
    assert all(torch.isfinite(v).all() for v in g8.values())
    assert rels[len(rels) // 2] <= 0.15 and rels[-1] <= 1.0, (rels[len(rels) // 2], rels[-3:])
