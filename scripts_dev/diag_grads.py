import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hri_emo_amd as H
from oracle import hri_emo_oracle as O
from conftest import load_golden
for name, d, ne in [("cfg1_train_p0", 128, 4), ("hd96_train_p0", 768, 6)]:
    g = load_golden(name)
    ref = O.closed_form_init_(O.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=0.0)).train()
    m = O.closed_form_init_(H.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=0.0)).cuda().train()
    ha, ht = g["h_a"].clone().requires_grad_(True), g["h_t"].clone().requires_grad_(True)
    l, b, z = ref(ha, ht, g["mask_a"], g["mask_t"]); O.train_step_loss(l, b, g["y"]).backward()
    ha2, ht2 = g["h_a"].cuda().requires_grad_(True), g["h_t"].cuda().requires_grad_(True)
    l2, b2, z2 = m(ha2, ht2, g["mask_a"].cuda(), g["mask_t"].cuda()); O.train_step_loss(l2, b2, g["y"].cuda()).backward()
    def rel(a, b): return ((a.cpu().float() - b).norm() / b.norm().clamp_min(1e-30)).item()
    print(name, "logits", rel(l2, l), "z", rel(z2, z), "beta", rel(b2, b))
    print("  g_h_a", rel(ha2.grad, ha.grad), "g_h_t", rel(ht2.grad, ht.grad))
    rows = []
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        rows.append((rel(p.grad, q.grad), n, q.grad.norm().item()))
    rows.sort(reverse=True)
    for r in rows[:12]: print("   %.4f  %-60s |g|=%.3e" % r)
    print("   median rel", sorted(r[0] for r in rows)[len(rows)//2])
