"""GPU suite, module level: the drop-in nn.Modules (hri_emo_amd.models.*) against
  (1) the golden vectors generated from the reference import (tests/golden/*.npz), and
  (2) the CPU oracle on seeded inputs,
plus size-independent properties at the full BASELINE cfg-2 size.

Tolerance (north_star: 1e-2 for the bf16 path): |got - ref| <= 1e-2 * max(1, max|ref|) for outputs;
gradients (bf16 backward through ~20 GEMMs) are held to 5e-2 of each tensor's max and 2e-2 on norms.
"""
import pytest
import torch

from conftest import load_golden
from oracle import hri_emo_oracle as O          # the checker (tests only)

pytestmark = pytest.mark.gpu
TOL = 1e-2


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
    for li, maps in enumerate(pack["encoder"]):
        for k, v in maps.items():
            close(v, g[f"enc.{li}.{k}"], what=f"enc.{li}.{k}")
    for li, v in enumerate(pack["decoder"]):
        close(v, g[f"dec.{li}"], what=f"dec.{li}")
    w = pack["encoder"][-1]["audio_queries_text"].cpu()
    assert (w[g["mask_t"][:, None, :].expand_as(w)] == 0).all()          # PAD key columns exactly 0
    close(w.sum(-1), torch.ones(w.shape[:-1]), 5e-3, "rows sum to one")


def test_fusion_allpad_row_nan_only_for_that_sample(H):
    g = load_golden("cfg1_eval_allpad_row")
    m = fusion(H, 128, 4).eval()
    with torch.no_grad():
        logits, beta, z = m(cu(g["h_a"]), cu(g["h_t"]), cu(g["mask_a"]), cu(g["mask_t"]))
    logits = logits.cpu()
    assert torch.equal(torch.isnan(logits), torch.isnan(g["logits"]))
    ok = ~torch.isnan(g["logits"])
    close(logits[ok], g["logits"][ok])


@pytest.mark.parametrize("name,d,ne", [("cfg1_train_p0", 128, 4), ("hd96_train_p0", 768, 6)])
def test_fusion_train_step_grads_vs_golden(H, name, d, ne):
    g = load_golden(name)
    m = fusion(H, d, ne, p=0.0).train()
    h_a = g["h_a"].cuda().requires_grad_(True)
    h_t = g["h_t"].cuda().requires_grad_(True)
    logits, beta, z = m(h_a, h_t, cu(g["mask_a"]), cu(g["mask_t"]))
    loss = O.train_step_loss(logits, beta, g["y"].cuda())
    loss.backward()
    close(loss.reshape(1), g["loss"], what="loss"); close(logits, g["logits"], what="logits")
    scale_a = g["g_h_a"].abs().max().item()
    assert (h_a.grad.cpu() - g["g_h_a"]).abs().max().item() <= 5e-2 * scale_a, "g_h_a"
    scale_t = g["g_h_t"].abs().max().item()
    assert (h_t.grad.cpu() - g["g_h_t"]).abs().max().item() <= 5e-2 * scale_t, "g_h_t"
    worst = []
    for n, p in m.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32, n
        gn = g["g.norm." + n].item()
        rel = abs(p.grad.norm().item() - gn) / max(gn, 1e-6)
        worst.append((rel, n))
        ref = g["g.full." + n] if "g.full." + n in g else None
        got = p.grad.cpu()
        if ref is None:
            flat = got.reshape(-1)
            idx = torch.linspace(0, flat.numel() - 1, 64).long()
            got, ref = flat[idx], g["g.samp." + n]
        bound = 5e-2 * max(ref.abs().max().item(), gn / max(1.0, p.numel() ** 0.5) * 4)
        assert (got - ref).abs().max().item() <= bound, (n, (got - ref).abs().max().item(), bound)
    worst.sort(reverse=True)
    assert worst[0][0] <= 2e-2, worst[:5]


def test_components_vs_golden(H):
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


def _rand_batch(B, Ta, Tt, d, seed, ragged=True):
    g = torch.Generator().manual_seed(seed)
    h_a, h_t = torch.randn(B, Ta, d, generator=g), torch.randn(B, Tt, d, generator=g)
    if not ragged:
        return h_a, h_t, None, None
    la = torch.randint(Ta // 2, Ta + 1, (B,), generator=g)
    lt = torch.randint(Tt // 2, Tt + 1, (B,), generator=g)
    return h_a, h_t, torch.arange(Ta)[None] >= la[:, None], torch.arange(Tt)[None] >= lt[:, None]


@pytest.mark.parametrize("B,Ta,Tt,d,ne", [(3, 100, 40, 768, 6), (2, 130, 50, 256, 7), (4, 32, 16, 128, 4)])
def test_fusion_vs_oracle_seeded(H, B, Ta, Tt, d, ne):
    torch.manual_seed(1234)
    ref = O.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=0.1)   # default torch init
    m = H.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=0.1)
    m.load_state_dict(ref.state_dict())
    m.cuda().eval(); ref.eval()
    h_a, h_t, m_a, m_t = _rand_batch(B, Ta, Tt, d, 5)
    with torch.no_grad():
        lr, br, zr = ref(h_a, h_t, m_a, m_t)
        lg, bg, zg = m(cu(h_a), cu(h_t), cu(m_a), cu(m_t))
    close(lg, lr, what="logits"); close(bg, br, what="beta"); close(zg, zr, what="z")


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
