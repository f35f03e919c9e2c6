"""CPU suite: the oracle (oracle/hri_emo_oracle.py) against the golden vectors generated from the
reference import (tests/golden/make_golden.py).  Tolerance: 1e-5 abs (fp32 restatement of fp32 math)."""
import pytest
import torch

from oracle import hri_emo_oracle as O
from conftest import load_golden

TOL = 1e-5


def close(a, b, tol=TOL):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert err <= tol * max(1.0, b.abs().max().item()), err


def _fusion(d, ne, p=0.1):
    return O.closed_form_init_(O.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=p))


def _check_maps(g, pack):
    for li, m in enumerate(pack["encoder"]):
        for k, v in m.items():
            close(v, g[f"enc.{li}.{k}"])
    for li, v in enumerate(pack["decoder"]):
        close(v, g[f"dec.{li}"])


def test_state_dict_keys_match_reference_key_list():
    # key names recorded in the train fixture (one "g.norm.<key>" per reference parameter)
    g = load_golden("cfg1_train_p0")
    ref_keys = [k[len("g.norm."):] for k in g if k.startswith("g.norm.")]
    mine = [n for n, _ in _fusion(128, 4).named_parameters()]
    assert mine == ref_keys


@pytest.mark.parametrize("name", ["cfg1_eval_nomask", "cfg1_eval_ragged", "cfg1_eval_2d_inputs"])
def test_fusion_eval(name):
    g = load_golden(name)
    m = _fusion(128, 4).eval()
    with torch.no_grad():
        logits, beta, z = m(g["h_a"], g["h_t"], g.get("mask_a"), g.get("mask_t"))
    close(logits, g["logits"]); close(beta, g["beta"]); close(z, g["z"])


def test_fusion_attention_maps():
    g = load_golden("cfg1_eval_ragged")
    m = _fusion(128, 4).eval()
    with torch.no_grad():
        logits, beta, z, pack = m(g["h_a"], g["h_t"], g["mask_a"], g["mask_t"], return_attention=True)
    close(logits, g["logits"])
    _check_maps(g, pack)
    # PAD key columns are exactly zero and rows sum to one (what the reference's notebook relies on)
    w = pack["encoder"][-1]["audio_queries_text"]
    assert (w[g["mask_t"][:, None, :].expand_as(w)] == 0).all()
    close(w.sum(-1), torch.ones_like(w.sum(-1)))


def test_fusion_allpad_row_is_nan_only_for_that_sample():
    g = load_golden("cfg1_eval_allpad_row")
    m = _fusion(128, 4).eval()
    with torch.no_grad():
        logits, beta, z = m(g["h_a"], g["h_t"], g["mask_a"], g["mask_t"])
    assert torch.equal(torch.isnan(logits), torch.isnan(g["logits"]))
    assert torch.isnan(logits[3]).all() and not torch.isnan(logits[[0, 1, 2, 4, 5, 6, 7]]).any()
    ok = ~torch.isnan(g["logits"])
    close(logits[ok], g["logits"][ok])


@pytest.mark.parametrize("name,d,ne", [("cfg1_train_p0", 128, 4), ("hd96_train_p0", 768, 6)])
def test_fusion_train_grads(name, d, ne):
    g = load_golden(name)
    m = _fusion(d, ne, p=0.0).train()
    h_a = g["h_a"].clone().requires_grad_(True)
    h_t = g["h_t"].clone().requires_grad_(True)
    logits, beta, z = m(h_a, h_t, g["mask_a"], g["mask_t"])
    loss = O.train_step_loss(logits, beta, g["y"])
    loss.backward()
    close(loss.reshape(1), g["loss"]); close(logits, g["logits"])
    close(h_a.grad, g["g_h_a"], 1e-4); close(h_t.grad, g["g_h_t"], 1e-4)
    for n, p in m.named_parameters():
        close(p.grad.norm().reshape(1), g["g.norm." + n], 1e-4)
        if "g.full." + n in g:
            close(p.grad, g["g.full." + n], 1e-4)
        else:
            flat = p.grad.reshape(-1)
            idx = torch.linspace(0, flat.numel() - 1, 64).long()
            close(flat[idx], g["g.samp." + n], 1e-4)
    if name == "cfg1_train_p0":
        # trainer step: clip 5.0 + AdamW(1e-4, wd 1e-2)  (train_fusion_seq_level_decoder.py:331-334)
        before = {n: p.detach().clone() for n, p in m.named_parameters()}
        opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-2)
        tn = torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
        opt.step()
        close(tn.reshape(1), g["total_grad_norm"], 1e-4)
        for n, p in m.named_parameters():
            close((p.detach() - before[n]).norm().reshape(1), g["delta.norm." + n], 1e-3)


def test_cfg2_shape_seeded_eval_and_train_step():
    """the oracle at the HEADLINE shape (d=768, T_a=400, T_t=128, N_e=6, B=2, ragged) against the reference's outputs, loss and
    gradient record"""
    from conftest import cfg2_seeded_inputs
    g = load_golden("cfg2_seeded")
    h_a, h_t, m_a, m_t = cfg2_seeded_inputs(g)
    m = _fusion(768, 6, p=0.0).eval()
    with torch.no_grad():
        logits, beta, z = m(h_a, h_t, m_a, m_t)
    close(logits, g["logits"]); close(beta, g["beta"]); close(z, g["z"])
    m.train()
    h_a = h_a.clone().requires_grad_(True)
    h_t = h_t.clone().requires_grad_(True)
    lt, bt, zt = m(h_a, h_t, m_a, m_t)
    loss = O.train_step_loss(lt, bt, g["y"])
    loss.backward()
    close(loss.reshape(1), g["loss"])
    close(h_a.grad.norm().reshape(1), g["g_h_a_norm"], 1e-4); close(h_t.grad.norm().reshape(1), g["g_h_t_norm"], 1e-4)
    close(h_a.grad.reshape(-1)[torch.linspace(0, h_a.numel() - 1, 256).long()], g["g_h_a_samp"], 1e-4)
    for n, p in m.named_parameters():
        close(p.grad.norm().reshape(1), g["g.norm." + n], 1e-4)
        if "g.full." + n in g:
            close(p.grad, g["g.full." + n], 1e-4)
        else:
            flat = p.grad.reshape(-1)
            close(flat[torch.linspace(0, flat.numel() - 1, 64).long()], g["g.samp." + n], 1e-4)


@pytest.mark.parametrize("name,seed,Ta,Tt,kw", [("cfg4_seeded", 31, 1000, 50, dict(d_model=768, num_emotions=6, n_heads=8)),
                                                ("cfg5_seeded", 41, 400, 128, dict(d_model=1024, num_emotions=7, n_heads=8,
                                                                                   num_layers_fusion=4, num_layers_decoder=2))])
def test_other_baseline_configs_seeded(name, seed, Ta, Tt, kw):
    """the oracle at BASELINE configs[3] (MOSEI shape) and at configs[4]'s dimensions against the reference's outputs"""
    from conftest import cfg2_seeded_inputs
    g = load_golden(name)
    h_a, h_t, m_a, m_t = cfg2_seeded_inputs(g, seed, Ta, Tt, kw["d_model"])
    m = O.closed_form_init_(O.FusionWithEmotionDecoder(dropout=0.0, **kw)).eval()
    with torch.no_grad():
        logits, beta, z = m(h_a, h_t, m_a, m_t)
    close(logits, g["logits"]); close(beta, g["beta"]); close(z, g["z"])
    close(O.train_step_loss(logits, beta, g["y"]).reshape(1), g["loss"])


def test_hd96_eval_and_maps():
    g = load_golden("hd96_eval_ragged")
    m = _fusion(768, 6, p=0.0).eval()
    with torch.no_grad():
        logits, beta, z, pack = m(g["h_a"], g["h_t"], g["mask_a"], g["mask_t"], return_attention=True)
    close(logits, g["logits"]); close(beta, g["beta"]); close(z, g["z"])
    _check_maps(g, pack)


def test_block_gate_decoder_components():
    g = load_golden("block_eval_ragged")
    blk = O.closed_form_init_(O.CrossModalBlock(128, 8, 0.1)).eval()
    with torch.no_grad():
        oa, ot, maps = blk(g["h_a"], g["h_t"], g["mask_a"], g["mask_t"], return_attention=True)
    close(oa, g["out_a"]); close(ot, g["out_t"])
    for k, v in maps.items():
        close(v, g["map." + k])

    gg = load_golden("gate_eval_ragged")
    gate = O.closed_form_init_(O.BetaGate(128, 32)).eval()
    with torch.no_grad():
        hf, beta = gate(gg["h_a"], gg["h_t"], gg["mask_a"], gg["mask_t"])
    close(hf, gg["h_fusion"]); close(beta, gg["beta"])
    gg = load_golden("gate_eval_equal_len_nomask")
    with torch.no_grad():
        hf, beta = gate(gg["h_a"], gg["h_t"])
    close(hf, gg["h_fusion"]); close(beta, gg["beta"])

    gd = load_golden("decoder_eval_ragged")
    dec = O.closed_form_init_(O.EmotionDecoder(128, 5, 8, 2, 64, 0.1)).eval()
    with torch.no_grad():
        z, logits, maps = dec(gd["memory"], gd["mask"], return_attention=True)
    close(z, gd["z"]); close(logits, gd["logits"])
    for i, v in enumerate(maps):
        close(v, gd[f"map.{i}"])


def test_mosei_wrapper():
    g = load_golden("mosei_eval_train")
    m = O.closed_form_init_(O.MoseiFusionWithEmotionDecoder(d_audio=74, d_text=300)).eval()
    with torch.no_grad():
        logits, beta, z = m(g["h_a"], g["h_t"], g["mask_a"], g["mask_t"])
    close(logits, g["logits"]); close(beta, g["beta"]); close(z, g["z"])
    mt = O.closed_form_init_(O.MoseiFusionWithEmotionDecoder(d_audio=74, d_text=300, dropout=0.0)).train()
    l2, b2, _ = mt(g["h_a"], g["h_t"], g["mask_a"], g["mask_t"])
    loss = O.train_step_loss(l2, b2, g["y"])
    loss.backward()
    close(loss.reshape(1), g["loss"])
    close(mt.audio_proj.weight.grad, g["g_audio_proj_w"], 1e-4); close(mt.text_proj.bias.grad, g["g_text_proj_b"], 1e-4)


def test_legacy_block_and_fusion_classifier():
    g = load_golden("legacy_eval")
    leg = O.closed_form_init_(O.LegacyCrossModalTransformer(2, 128, 8, 0.1)).eval()
    with torch.no_grad():
        oa, ot = leg(g["h_a"], g["h_t"], g["mask_a"], g["mask_t"])
    close(oa, g["leg_a"]); close(ot, g["leg_t"])
    clf = O.closed_form_init_(O.FusionClassifier(128, 4, 8, 2, 32)).eval()
    with torch.no_grad():
        logits, beta, pooled = clf(g["h_a"], g["h_t"], g["mask_a"], g["mask_t"])
        l2, b2, p2 = clf(g["u_a"], g["u_t"])
    close(logits, g["clf_logits"]); close(beta, g["clf_beta"]); close(pooled, g["clf_pooled"])
    close(l2, g["u_logits"]); close(p2, g["u_pooled"])


def test_legacy_scalar_beta_gate():
    """models/beta_gate.py (legacy scalar gate): forward, and gradients w.r.t. inputs and MLP."""
    g = load_golden("legacy_gate")
    gate = O.closed_form_init_(O.LegacyBetaGate(128, 32))
    h_a, h_t = g["h_a"].clone().requires_grad_(True), g["h_t"].clone().requires_grad_(True)
    hf, beta = gate(h_a, h_t, g["mask_a"], g["mask_t"])
    close(hf, g["h_fusion"]); close(beta, g["beta"])
    ((hf * g["c"]).sum() + 3.0 * beta.sum()).backward()
    close(h_a.grad, g["g_h_a"], 1e-4); close(h_t.grad, g["g_h_t"], 1e-4)
    close(gate.mlp[0].weight.grad, g["g_w1"], 1e-4); close(gate.mlp[2].bias.grad, g["g_b2"], 1e-4)
    with torch.no_grad():
        hu, bu = gate(g["u_a"], g["u_t"])
    close(hu, g["u_h"]); close(bu, g["u_beta"])
