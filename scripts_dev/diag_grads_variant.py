"""per-parameter gradient errors of the cfg1 closed-form fixture with the chunk-mapped (1) and quad-mapped (0) add_ln kernels"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hri_emo_amd as H
from hri_emo_amd import _lib
from oracle import hri_emo_oracle as O
from conftest import load_golden
name, d, ne = "cfg1_train_p0", 128, 4
g = load_golden(name)
ref = O.closed_form_init_(O.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=0.0)).train()
ha, ht = g["h_a"].clone().requires_grad_(True), g["h_t"].clone().requires_grad_(True)
l, b, z = ref(ha, ht, g["mask_a"], g["mask_t"]); O.train_step_loss(l, b, g["y"]).backward()
def rel(a, b): return ((a.cpu().float() - b).norm() / b.norm().clamp_min(1e-30)).item()
res = {}
for v in (1, 0):
    _lib.call("hriemo_rowops_force_variant", v)
    m = O.closed_form_init_(H.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=0.0)).cuda().train()
    ha2, ht2 = g["h_a"].cuda().requires_grad_(True), g["h_t"].cuda().requires_grad_(True)
    l2, b2, z2 = m(ha2, ht2, g["mask_a"].cuda(), g["mask_t"].cuda()); O.train_step_loss(l2, b2, g["y"].cuda()).backward()
    torch.cuda.synchronize()
    print("variant", v, "logits", rel(l2, l), "z", rel(z2, z), "g_h_a", rel(ha2.grad, ha.grad), "g_h_t", rel(ht2.grad, ht.grad))
    res[v] = {n: (rel(p.grad, q.grad), p.grad.detach().cpu().clone()) for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters())}
for n in res[0]:
    e1, e0 = res[1][n][0], res[0][n][0]
    dd = rel(res[0][n][1], res[1][n][1])
    flag = " <<<" if e0 > 1.3 * e1 + 0.005 else ""
    print("  %-58s chunk %.4f quad %.4f  quad-vs-chunk %.4f%s" % (n, e1, e0, dd, flag))
