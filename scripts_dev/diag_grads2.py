import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hri_emo_amd as H
from oracle import hri_emo_oracle as O
from conftest import load_golden
def rel(a, b): return ((a.cpu().float() - b).norm() / b.norm().clamp_min(1e-30)).item()
for init in ["closed", "random"]:
  for name, d, ne in [("cfg1_train_p0", 128, 4), ("hd96_train_p0", 768, 6)]:
    g = load_golden(name)
    torch.manual_seed(1234)
    ref = O.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=0.0).train()
    if init == "closed": O.closed_form_init_(ref)
    m = H.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=0.0)
    m.load_state_dict(ref.state_dict()); m.cuda().train()
    ha, ht = g["h_a"].clone().requires_grad_(True), g["h_t"].clone().requires_grad_(True)
    l, b, z = ref(ha, ht, g["mask_a"], g["mask_t"]); O.train_step_loss(l, b, g["y"]).backward()
    gref = {n: p.grad.clone() for n, p in ref.named_parameters()}
    gha = ha.grad.clone()
    # autocast-bf16 oracle on CPU (yardstick for what bf16 arithmetic costs on these weights)
    ref.zero_grad(); ha.grad = None; ht.grad = None
    with torch.autocast("cpu", dtype=torch.bfloat16):
        l3, b3, z3 = ref(ha, ht, g["mask_a"], g["mask_t"])
    O.train_step_loss(l3.float(), b3.float(), g["y"]).backward()
    ac = sorted(rel(p.grad, gref[n]) for n, p in ref.named_parameters())
    ha2, ht2 = g["h_a"].cuda().requires_grad_(True), g["h_t"].cuda().requires_grad_(True)
    l2, b2, z2 = m(ha2, ht2, g["mask_a"].cuda(), g["mask_t"].cuda()); O.train_step_loss(l2, b2, g["y"].cuda()).backward()
    mine = sorted(rel(p.grad, gref[n]) for n, p in m.named_parameters())
    worst = sorted(((rel(p.grad, gref[n]), n) for n, p in m.named_parameters()), reverse=True)[:4]
    print("      worst:", [(round(a, 3), b) for a, b in worst])
    print(f"{init:7s} {name:14s} fwd z: mine {rel(z2, z):.4f} autocast {rel(z3, z):.4f} | param-grad rel err median/max: mine {mine[len(mine)//2]:.4f}/{mine[-1]:.4f}  autocast-bf16 {ac[len(ac)//2]:.4f}/{ac[-1]:.4f} | g_h_a mine {rel(ha2.grad, gha):.4f} autocast {rel(ha.grad, gha):.4f}")
