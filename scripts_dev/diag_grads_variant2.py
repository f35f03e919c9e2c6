"""quad vs chunk add_ln kernels: gradient difference between the two variants and error vs the fp32 oracle, closed-form vs default init"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hri_emo_amd as H
from hri_emo_amd import _lib
from oracle import hri_emo_oracle as O
from conftest import load_golden
def rel(a, b): return ((a.cpu().float() - b.cpu().float()).norm() / b.cpu().float().norm().clamp_min(1e-30)).item()
for name, d, ne in (("cfg1_train_p0", 128, 4), ("hd96_train_p0", 768, 6)):
    g = load_golden(name)
    for init in ("closed", "random"):
        torch.manual_seed(1234)
        ref = O.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=0.0).train()
        if init == "closed":
            O.closed_form_init_(ref)
        ha, ht = g["h_a"].clone().requires_grad_(True), g["h_t"].clone().requires_grad_(True)
        l, b, z = ref(ha, ht, g["mask_a"], g["mask_t"]); O.train_step_loss(l, b, g["y"]).backward()
        res = {}
        for v in (1, 0):
            _lib.call("hriemo_rowops_force_variant", v)
            m = H.FusionWithEmotionDecoder(d_model=d, num_emotions=ne, n_heads=8, dropout=0.0)
            m.load_state_dict(ref.state_dict()); m.cuda().train()
            ha2, ht2 = g["h_a"].cuda().requires_grad_(True), g["h_t"].cuda().requires_grad_(True)
            l2, b2, z2 = m(ha2, ht2, g["mask_a"].cuda(), g["mask_t"].cuda()); O.train_step_loss(l2, b2, g["y"].cuda()).backward()
            torch.cuda.synchronize()
            res[v] = ({n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}, z2.detach().cpu(), ha2.grad.cpu())
        gr = {n: p.grad for n, p in ref.named_parameters()}
        e1 = sorted(rel(res[1][0][n], gr[n]) for n in gr); e0 = sorted(rel(res[0][0][n], gr[n]) for n in gr)
        dd = sorted(rel(res[0][0][n], res[1][0][n]) for n in gr)
        k = len(gr)
        print(f"{name}/{init}: err vs oracle median chunk {e1[k//2]:.4f} quad {e0[k//2]:.4f}; p90 chunk {e1[k*9//10]:.4f} quad {e0[k*9//10]:.4f}; "
              f"quad-vs-chunk median {dd[k//2]:.4f} max {dd[-1]:.4f}; z diff {rel(res[0][1], res[1][1]):.2e}; dh_a diff {rel(res[0][2], res[1][2]):.4f}", flush=True)
_lib.call("hriemo_rowops_force_variant", 1)
