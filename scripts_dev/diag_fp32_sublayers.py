"""fp32 mode: each piece of the backward against float64 on small / large row counts"""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd as H
from hri_emo_amd import _fp32, _ops
H.set_precision("fp32")
def rel(a, b): return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-300)).item()
g = torch.Generator().manual_seed(3)
sh = _ops.Shadows()
for M, N, K in ((40, 1024, 4096), (40, 4096, 1024), (96, 768, 3072), (2, 768, 256), (2, 256, 3072), (1000, 768, 768)):
    dy = torch.randn(M, N, generator=g); w = torch.nn.Parameter((torch.randn(N, K, generator=g) / math.sqrt(K)).cuda()); x = torch.randn(M, K, generator=g)
    mk = torch.randn(M, N, generator=g)
    dx = _fp32.linear_dx(dy.cuda(), sh, w); r = dy.double() @ w.detach().double().cpu()
    dxm = _fp32.linear_dx(dy.cuda(), sh, w, mask=mk.cuda()); rm = (dy.double() * (mk > 0)) @ w.detach().double().cpu()
    base = torch.randn(M, K, generator=g)
    dxi = _fp32.linear_dx(dy.cuda(), sh, w, into=base.clone().cuda()); ri = r + base.double()
    dw = _fp32.linear_dw(dy.cuda(), x.cuda()); rw = dy.double().t() @ x.double()
    dwm = _fp32.linear_dw(dy.cuda(), x.cuda(), mask=mk.cuda()); rwm = (dy.double() * (mk > 0)).t() @ x.double()
    dwr = _fp32.linear_dw(dy.cuda(), x.cuda(), relu_x=True); rwr = dy.double().t() @ x.double().clamp_min(0)
    y = _fp32.linear(x.cuda(), sh, w, None); ry = x.double() @ w.detach().double().cpu().t()
    print(f"M{M} N{N} K{K}: fwd {rel(y, ry):.1e} dx {rel(dx, r):.1e} dx_mask {rel(dxm, rm):.1e} dx_into {rel(dxi, ri):.1e} dw {rel(dw, rw):.1e} dw_mask {rel(dwm, rwm):.1e} dw_relu {rel(dwr, rwr):.1e}", flush=True)
