"""fp32 training step: per-parameter gradient error of (a) the fp32 oracle and (b) the fp32 mode against the float64 oracle"""
import sys, os, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hri_emo_amd as H
from oracle import hri_emo_oracle as O
from conftest import load_golden
H.set_precision("fp32")
def rel(a, b): return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-300)).item()
def step(model, h_a, h_t, m_a, m_t, y):
    h_a = h_a.clone().requires_grad_(True); h_t = h_t.clone().requires_grad_(True)
    logits, beta, z = model(h_a, h_t, m_a, m_t)
    loss = O.train_step_loss(logits, beta, y)
    model.zero_grad(); loss.backward()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters()}, h_a.grad, h_t.grad
def rand_batch(B, Ta, Tt, d, seed):
    g = torch.Generator().manual_seed(seed)
    h_a, h_t = torch.randn(B, Ta, d, generator=g), torch.randn(B, Tt, d, generator=g)
    la = torch.randint(max(1, Ta // 2), Ta + 1, (B,), generator=g); lt = torch.randint(max(1, Tt // 2), Tt + 1, (B,), generator=g)
    return h_a, h_t, torch.arange(Ta)[None] >= la[:, None], torch.arange(Tt)[None] >= lt[:, None]
cases = []
g = load_golden("hd96_train_p0")
cases.append(("hd96_train_p0 closed", dict(d_model=768, num_emotions=6, n_heads=8, dropout=0.0), True, (g["h_a"], g["h_t"], g["mask_a"], g["mask_t"], g["y"])))
kw = dict(d_model=1024, num_emotions=7, n_heads=8, num_layers_fusion=1, num_layers_decoder=1, dropout=0.0)
h_a, h_t, m_a, m_t = rand_batch(2, 90, 20, 1024, 79)
y = (torch.rand(2, 7, generator=torch.Generator().manual_seed(9)) < 0.3).float()
cases.append(("d1024 default init", kw, False, (h_a, h_t, m_a, m_t, y)))
for name, kw, closed, (h_a, h_t, m_a, m_t, y) in cases:
    torch.manual_seed(1234)
    ref = O.FusionWithEmotionDecoder(**kw).train()
    if closed:
        O.closed_form_init_(ref)
    ref64 = copy.deepcopy(ref).double()
    m = H.FusionWithEmotionDecoder(**kw); m.load_state_dict(ref.state_dict()); m.cuda().train()
    g64, a64, t64 = step(ref64, h_a.double(), h_t.double(), m_a, m_t, y.double())
    g32, a32, t32 = step(ref, h_a, h_t, m_a, m_t, y)
    gm, am, tm = step(m, h_a.cuda(), h_t.cuda(), m_a.cuda(), m_t.cuda(), y.cuda())
    rows = sorted(((rel(gm[n], g64[n]), rel(g32[n], g64[n]), rel(gm[n], g32[n]), n) for n in g64), reverse=True)
    print(name, "inputs: mine-vs-64", rel(am, a64), rel(tm, t64), "oracle32-vs-64", rel(a32, a64), rel(t32, t64))
    for r in rows[:10]:
        print("   mine-vs-f64 %.2e  oracle32-vs-f64 %.2e  mine-vs-oracle32 %.2e  %s" % r)
    print("   median mine %.2e oracle32 %.2e" % (sorted(r[0] for r in rows)[len(rows)//2], sorted(r[1] for r in rows)[len(rows)//2]), flush=True)
