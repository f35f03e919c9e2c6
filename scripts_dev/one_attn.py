"""Run the self_a attention kernels a few times (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd
from hri_emo_amd import _ops
B, H, hd, Lq, Lk, p = 64, 8, 96, 400, 400, 0.1
d = H * hd
q = torch.randn(B * Lq, d, device="cuda").bfloat16(); k = torch.randn(B * Lk, d, device="cuda").bfloat16(); v = torch.randn(B * Lk, d, device="cuda").bfloat16()
o, lse = _ops.attn_fwd(q, k, v, B, H, Lq, Lk, hd, None, p, 1234, 5, 0)
do = torch.randn_like(o); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
for _ in range(3):
    _ops.attn_fwd(q, k, v, B, H, Lq, Lk, hd, None, p, 1234, 5, 0)
    _ops.attn_bwd(q, k, v, o, do, dq, dk, dv, lse, B, H, Lq, Lk, hd, None, p, 1234, 5, 0)
torch.cuda.synchronize()
