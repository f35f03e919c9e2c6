#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference's own
``models/*`` (read-only tree at /root/reference) on CPU/fp32 in the build container.

Only inputs and expected outputs are written (``*.npz``); no reference source travels.
Weights are never stored: both sides fill parameters with the closed-form initialiser
``oracle.hri_emo_oracle.closed_form_init_`` (same ``named_parameters()`` order/keys).

Run (build container only -- /root/reference does not exist on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

from oracle.hri_emo_oracle import closed_form_init_, train_step_loss  # noqa: E402
from models.fusion_with_emotion_decoder import FusionWithEmotionDecoder  # noqa: E402  (reference)
from models.cross_modal_block_tacfn import CrossModalBlock  # noqa: E402  (reference)
from models.beta_gate_tacfn import BetaGate  # noqa: E402  (reference)
from models.emotion_decoder import EmotionDecoder  # noqa: E402  (reference)
from models.mosei_fusion_with_emotion_decoder import MoseiFusionWithEmotionDecoder  # noqa: E402  (reference)

torch.set_num_threads(4)


def inputs(seed, B, Ta, Tt, d, ragged):
    g = torch.Generator().manual_seed(seed)
    h_a = torch.randn(B, Ta, d, generator=g)
    h_t = torch.randn(B, Tt, d, generator=g)
    if not ragged:
        return h_a, h_t, None, None
    la = torch.randint(max(1, Ta // 2), Ta + 1, (B,), generator=g)
    lt = torch.randint(max(1, Tt // 2), Tt + 1, (B,), generator=g)
    la[0], lt[0] = Ta, Tt                      # one full-length sample
    m_a = torch.arange(Ta)[None, :] >= la[:, None]
    m_t = torch.arange(Tt)[None, :] >= lt[:, None]
    return h_a, h_t, m_a, m_t


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if v is None:
            continue
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, keys={len(out)}")


def grad_record(model, prefix="g."):
    """Full grads for 1-D params and emotion_queries; L2 norm + 64 strided samples otherwise."""
    rec = {}
    for name, p in model.named_parameters():
        g = p.grad
        rec[prefix + "norm." + name] = g.norm().reshape(1)
        if g.dim() <= 1 or name.endswith("emotion_queries"):
            rec[prefix + "full." + name] = g.clone()
        else:
            flat = g.reshape(-1)
            idx = torch.linspace(0, flat.numel() - 1, 64).long()
            rec[prefix + "samp." + name] = flat[idx].clone()
    return rec


def attn_record(pack):
    rec = {}
    for li, m in enumerate(pack["encoder"]):
        for k, v in m.items():
            rec[f"enc.{li}.{k}"] = v
    for li, v in enumerate(pack["decoder"]):
        rec[f"dec.{li}"] = v
    return rec


def fusion_cases():
    cfg1 = dict(d_model=128, num_emotions=4, n_heads=8)
    B, Ta, Tt, d = 8, 32, 16, 128

    # (1) cfg 1 eval, no masks  (BASELINE.json configs[0])
    m = closed_form_init_(FusionWithEmotionDecoder(**cfg1)).eval()
    h_a, h_t, _, _ = inputs(11, B, Ta, Tt, d, False)
    with torch.no_grad():
        logits, beta, z = m(h_a, h_t)
    npz("cfg1_eval_nomask", h_a=h_a, h_t=h_t, logits=logits, beta=beta, z=z)

    # (2)+(3) cfg 1 eval, ragged masks, attention maps
    h_a, h_t, m_a, m_t = inputs(12, B, Ta, Tt, d, True)
    with torch.no_grad():
        logits, beta, z = m(h_a, h_t, m_a, m_t)
        l2, b2, z2, pack = m(h_a, h_t, m_a, m_t, return_attention=True)
    assert torch.allclose(logits, l2, atol=1e-5)
    npz("cfg1_eval_ragged", h_a=h_a, h_t=h_t, mask_a=m_a, mask_t=m_t, logits=logits, beta=beta, z=z,
        **attn_record(pack))

    # (6) 2-D inputs -> _ensure_3d
    g = torch.Generator().manual_seed(13)
    xa, xt = torch.randn(B, d, generator=g), torch.randn(B, d, generator=g)
    with torch.no_grad():
        logits, beta, z = m(xa, xt)
    npz("cfg1_eval_2d_inputs", h_a=xa, h_t=xt, logits=logits, beta=beta, z=z)

    # (7) one sample whose text keys are all PAD -> NaN for that sample only
    h_a, h_t, m_a, m_t = inputs(14, B, Ta, Tt, d, True)
    m_t[3, :] = True
    with torch.no_grad():
        logits, beta, z = m(h_a, h_t, m_a, m_t)
    npz("cfg1_eval_allpad_row", h_a=h_a, h_t=h_t, mask_a=m_a, mask_t=m_t, logits=logits, beta=beta, z=z)

    # (4) train mode, dropout=0, loss of the seq-level trainer, grads; then clip + AdamW step
    mt = closed_form_init_(FusionWithEmotionDecoder(dropout=0.0, **cfg1)).train()
    h_a, h_t, m_a, m_t = inputs(15, B, Ta, Tt, d, True)
    h_a.requires_grad_(True)
    h_t.requires_grad_(True)
    y = (torch.rand(B, 4, generator=torch.Generator().manual_seed(16)) < 0.3).float()
    logits, beta, z = mt(h_a, h_t, m_a, m_t)
    loss = train_step_loss(logits, beta, y)
    loss.backward()
    rec = grad_record(mt)
    before = {n: p.detach().clone() for n, p in mt.named_parameters()}
    opt = torch.optim.AdamW(mt.parameters(), lr=1e-4, weight_decay=1e-2)
    total_norm = torch.nn.utils.clip_grad_norm_(mt.parameters(), max_norm=5.0)
    opt.step()
    for n, p in mt.named_parameters():
        rec["delta.norm." + n] = (p.detach() - before[n]).norm().reshape(1)
    npz("cfg1_train_p0", h_a=h_a, h_t=h_t, mask_a=m_a, mask_t=m_t, y=y, logits=logits, beta=beta, z=z,
        loss=loss.reshape(1), total_grad_norm=total_norm.reshape(1), g_h_a=h_a.grad, g_h_t=h_t.grad, **rec)  # rec holds clones taken BEFORE the clip

    # (5) real head_dim: d=768, H=8 (hd=96), lengths not multiples of 16
    cfg = dict(d_model=768, num_emotions=6, n_heads=8)
    B2, Ta2, Tt2 = 2, 48, 20
    m768 = closed_form_init_(FusionWithEmotionDecoder(dropout=0.0, **cfg)).eval()
    h_a, h_t, m_a, m_t = inputs(17, B2, Ta2, Tt2, 768, True)
    with torch.no_grad():
        logits, beta, z, pack = m768(h_a, h_t, m_a, m_t, return_attention=True)
    npz("hd96_eval_ragged", h_a=h_a, h_t=h_t, mask_a=m_a, mask_t=m_t, logits=logits, beta=beta, z=z,
        **attn_record(pack))
    m768.train()
    y = (torch.rand(B2, 6, generator=torch.Generator().manual_seed(18)) < 0.3).float()
    h_a.requires_grad_(True)
    h_t.requires_grad_(True)
    logits, beta, z = m768(h_a, h_t, m_a, m_t)
    loss = train_step_loss(logits, beta, y)
    loss.backward()
    npz("hd96_train_p0", h_a=h_a, h_t=h_t, mask_a=m_a, mask_t=m_t, y=y, logits=logits, beta=beta, z=z,
        loss=loss.reshape(1), g_h_a=h_a.grad, g_h_t=h_t.grad, **grad_record(m768))


def cfg2_seeded_case():
    """The headline shape itself (BASELINE.json configs[1]: d=768, T_a=400, T_t=128, N_e=6, H=8), B=2, ragged masks.  The inputs
    are NOT stored (1.6 MB of noise): the tests regenerate them from the same seeded generator and check the stored probes
    (first values, sums) before using them; outputs, loss and the gradient record are stored."""
    cfg = dict(d_model=768, num_emotions=6, n_heads=8)
    B, Ta, Tt, d = 2, 400, 128, 768
    m = closed_form_init_(FusionWithEmotionDecoder(dropout=0.0, **cfg)).eval()
    h_a, h_t, m_a, m_t = inputs(21, B, Ta, Tt, d, True)
    probe = torch.cat([h_a[0, 0, :8], h_t[1, -1, -8:], h_a.sum().reshape(1), h_t.sum().reshape(1)])
    with torch.no_grad():
        logits, beta, z = m(h_a, h_t, m_a, m_t)
    m.train()
    y = (torch.rand(B, 6, generator=torch.Generator().manual_seed(22)) < 0.3).float()
    h_a.requires_grad_(True)
    h_t.requires_grad_(True)
    lt, bt, zt = m(h_a, h_t, m_a, m_t)
    loss = train_step_loss(lt, bt, y)
    loss.backward()
    npz("cfg2_seeded", probe=probe, mask_a=m_a, mask_t=m_t, y=y, logits=logits, beta=beta, z=z, loss=loss.reshape(1),
        g_h_a_norm=h_a.grad.norm().reshape(1), g_h_t_norm=h_t.grad.norm().reshape(1),
        g_h_a_samp=h_a.grad.reshape(-1)[torch.linspace(0, h_a.numel() - 1, 256).long()],
        g_h_t_samp=h_t.grad.reshape(-1)[torch.linspace(0, h_t.numel() - 1, 256).long()], **grad_record(m))


def other_config_cases():
    """BASELINE.json configs[3] (MOSEI shape d=768, T_a=1000, T_t=50, N_e=6) and configs[4]'s dimensions (d=1024, 4 fusion + 2
    decoder layers, N_e=7, T_a=400, T_t=128), B=2, ragged masks: outputs and the trainer loss only, inputs regenerated from the
    seed like cfg2_seeded."""
    for name, seed, cfg, (Ta, Tt) in (("cfg4_seeded", 31, dict(d_model=768, num_emotions=6, n_heads=8), (1000, 50)),
                                      ("cfg5_seeded", 41, dict(d_model=1024, num_emotions=7, n_heads=8, num_layers_fusion=4,
                                                               num_layers_decoder=2), (400, 128))):
        B, d = 2, cfg["d_model"]
        m = closed_form_init_(FusionWithEmotionDecoder(dropout=0.0, **cfg)).eval()
        h_a, h_t, m_a, m_t = inputs(seed, B, Ta, Tt, d, True)
        probe = torch.cat([h_a[0, 0, :8], h_t[1, -1, -8:], h_a.sum().reshape(1), h_t.sum().reshape(1)])
        with torch.no_grad():
            logits, beta, z = m(h_a, h_t, m_a, m_t)
        y = (torch.rand(B, cfg["num_emotions"], generator=torch.Generator().manual_seed(seed + 1)) < 0.3).float()
        loss = train_step_loss(logits, beta, y)
        npz(name, probe=probe, mask_a=m_a, mask_t=m_t, y=y, logits=logits, beta=beta, z=z, loss=loss.reshape(1))


def component_cases():
    d, H, B = 128, 8, 4
    # CrossModalBlock alone (a1-a5)
    blk = closed_form_init_(CrossModalBlock(d_model=d, n_heads=H, dropout=0.1)).eval()
    h_a, h_t, m_a, m_t = inputs(21, B, 24, 10, d, True)
    with torch.no_grad():
        oa, ot, maps = blk(h_a, h_t, m_a, m_t, return_attention=True)
    npz("block_eval_ragged", h_a=h_a, h_t=h_t, mask_a=m_a, mask_t=m_t, out_a=oa, out_t=ot,
        **{"map." + k: v for k, v in maps.items()})

    # BetaGate alone: L_a != L_t (truncate to L_t) and L_a == L_t (a7, a8)
    gate = closed_form_init_(BetaGate(d_model=d, hidden_dim=32)).eval()
    with torch.no_grad():
        hf, beta = gate(h_a, h_t, m_a, m_t)
    npz("gate_eval_ragged", h_a=h_a, h_t=h_t, mask_a=m_a, mask_t=m_t, h_fusion=hf, beta=beta)
    h_a2, h_t2, _, _ = inputs(22, B, 12, 12, d, False)
    with torch.no_grad():
        hf, beta = gate(h_a2, h_t2)
    npz("gate_eval_equal_len_nomask", h_a=h_a2, h_t=h_t2, h_fusion=hf, beta=beta)

    # EmotionDecoder alone (a9, a10)
    dec = closed_form_init_(EmotionDecoder(d_model=d, num_emotions=5, n_heads=H, num_layers=2,
                                           dim_feedforward=64, dropout=0.1)).eval()
    mem = h_t
    with torch.no_grad():
        z, logits, maps = dec(mem, m_t, return_attention=True)
    npz("decoder_eval_ragged", memory=mem, mask=m_t, z=z, logits=logits,
        **{f"map.{i}": v for i, v in enumerate(maps)})


def mosei_case():
    """COVAREP d=74 / GloVe d=300 projections in front of the backbone (MOSEI defaults)."""
    m = closed_form_init_(MoseiFusionWithEmotionDecoder(d_audio=74, d_text=300)).eval()
    g = torch.Generator().manual_seed(31)
    B, Ta, Tt = 3, 50, 20
    h_a, h_t = torch.randn(B, Ta, 74, generator=g), torch.randn(B, Tt, 300, generator=g)
    la = torch.randint(Ta // 2, Ta + 1, (B,), generator=g)
    lt = torch.randint(Tt // 2, Tt + 1, (B,), generator=g)
    m_a, m_t = torch.arange(Ta)[None] >= la[:, None], torch.arange(Tt)[None] >= lt[:, None]
    with torch.no_grad():
        logits, beta, z = m(h_a, h_t, m_a, m_t)
    mt = closed_form_init_(MoseiFusionWithEmotionDecoder(d_audio=74, d_text=300, dropout=0.0)).train()
    y = (torch.rand(B, 6, generator=g) < 0.3).float()
    l2, b2, _ = mt(h_a, h_t, m_a, m_t)
    loss = train_step_loss(l2, b2, y)
    loss.backward()
    npz("mosei_eval_train", h_a=h_a, h_t=h_t, mask_a=m_a, mask_t=m_t, logits=logits, beta=beta, z=z, y=y,
        loss=loss.reshape(1), g_audio_proj_w=mt.audio_proj.weight.grad, g_audio_proj_b=mt.audio_proj.bias.grad,
        g_text_proj_w=mt.text_proj.weight.grad, g_text_proj_b=mt.text_proj.bias.grad)


def legacy_cases():
    """legacy cross-modal block (no self-attention) and FusionClassifier."""
    from models.cross_modal_block import CrossModalTransformer as LegacyCMT   # reference
    from models.fusion_classifier import FusionClassifier                      # reference
    d, B = 128, 4
    h_a, h_t, m_a, m_t = inputs(41, B, 24, 10, d, True)
    leg = closed_form_init_(LegacyCMT(num_layers=2, d_model=d, n_heads=8, dropout=0.1)).eval()
    with torch.no_grad():
        oa, ot = leg(h_a, h_t, m_a, m_t)
    clf = closed_form_init_(FusionClassifier(d_model=d, num_classes=4, n_heads=8, num_layers=2, beta_hidden=32)).eval()
    with torch.no_grad():
        logits, beta, pooled = clf(h_a, h_t, m_a, m_t)
        g = torch.Generator().manual_seed(42)
        xa, xt = torch.randn(B, d, generator=g), torch.randn(B, d, generator=g)
        l2, b2, p2 = clf(xa, xt)                                               # utterance-level [B, d] inputs
    npz("legacy_eval", h_a=h_a, h_t=h_t, mask_a=m_a, mask_t=m_t, leg_a=oa, leg_t=ot, clf_logits=logits, clf_beta=beta,
        clf_pooled=pooled, u_a=xa, u_t=xt, u_logits=l2, u_beta=b2, u_pooled=p2)


def legacy_gate_case():
    """the legacy scalar gate models/beta_gate.py (reference tests/test_beta_gate.py), sequence
    inputs with ragged masks and unequal lengths (fusion length = text length) and the test's own [B,1,d] shape;
    loss = sum(h_fusion * c) + 3 * sum(beta) with a fixed cotangent c, gradients w.r.t. inputs and MLP."""
    from models.beta_gate import BetaGate as LegacyGate                       # reference
    d, B = 128, 4
    h_a, h_t, m_a, m_t = inputs(51, B, 24, 10, d, True)
    gate = closed_form_init_(LegacyGate(d_model=d, hidden_dim=32))
    g = torch.Generator().manual_seed(52)
    c = torch.randn(B, 10, d, generator=g)
    h_a.requires_grad_(True); h_t.requires_grad_(True)
    hf, beta = gate(h_a, h_t, m_a, m_t)
    ((hf * c).sum() + 3.0 * beta.sum()).backward()
    xa, xt = torch.randn(B, 1, d, generator=g), torch.randn(B, 1, d, generator=g)
    with torch.no_grad():
        hu, bu = gate(xa, xt)
    npz("legacy_gate", h_a=h_a.detach(), h_t=h_t.detach(), mask_a=m_a, mask_t=m_t, c=c, h_fusion=hf.detach(), beta=beta.detach(),
        g_h_a=h_a.grad, g_h_t=h_t.grad, g_w1=gate.mlp[0].weight.grad, g_b1=gate.mlp[0].bias.grad,
        g_w2=gate.mlp[2].weight.grad, g_b2=gate.mlp[2].bias.grad, u_a=xa, u_t=xt, u_h=hu, u_beta=bu)


def collate_case():
    """the trainer's collate (scripts/fusion/train_fusion_seq_level_decoder.py:191-232) on ragged
    samples whose stored masks already contain PAD tails."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_trainer", "/root/reference/scripts/fusion/train_fusion_seq_level_decoder.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                                               # reference (defines functions only)
    g = torch.Generator().manual_seed(61)
    d, C = 16, 4
    las, lts, vas, vts = [9, 14, 6, 11], [5, 3, 7, 4], [7, 14, 2, 8], [5, 2, 6, 1]
    batch, rec = [], {}
    for i, (la, lt, va, vt) in enumerate(zip(las, lts, vas, vts)):
        xa, xt = torch.randn(la, d, generator=g), torch.randn(lt, d, generator=g)
        ka, kt = torch.arange(la) >= va, torch.arange(lt) >= vt
        y = torch.zeros(C); y[i % C] = 1.0
        batch.append((xa, ka, xt, kt, y))
        rec.update({f"xa{i}": xa, f"ka{i}": ka, f"xt{i}": xt, f"kt{i}": kt, f"y{i}": y})
    h_a, m_a, h_t, m_t, labels = mod.collate_seq_batch(batch, "multi_label")
    _, _, _, _, single = mod.collate_seq_batch([(b[0], b[1], b[2], b[3], int(i % C)) for i, b in enumerate(batch)], "single_label")
    npz("collate", h_a=h_a, mask_a=m_a, h_t=h_t, mask_t=m_t, labels=labels, single=single, **rec)


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "collate":
        collate_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "legacy_gate":      # add this fixture without rewriting the others
        legacy_gate_case()
        sys.exit(0)
    only = sys.argv[1] if len(sys.argv) > 1 else None
    if only == "cfg2_seeded":
        cfg2_seeded_case()
        sys.exit(0)
    if only == "other_configs":
        other_config_cases()
        sys.exit(0)
    fusion_cases()
    cfg2_seeded_case()
    other_config_cases()
    component_cases()
    mosei_case()
    legacy_cases()
    legacy_gate_case()
    collate_case()
