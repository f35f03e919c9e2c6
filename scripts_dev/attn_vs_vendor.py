"""Yardstick only (never on the product path): the hand-written attention cores next to torch's
scaled_dot_product_attention (whatever fused backend this ROCm build ships) on the attention shapes of cfg 2."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd
from hri_emo_amd import _ops
dev = "cuda"
B, H, hd, pd = 64, 8, 96, 0.1

def t_us(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

print("site                  Lq   Lk |  mine fwd / bwd us (TF fwd+bwd)   vendor fwd / bwd us (TF fwd+bwd)")
for name, Lq, Lk in (("audio_queries_text", 400, 128), ("text_queries_audio", 128, 400), ("self_audio", 400, 400), ("self_text", 128, 128)):
    d = H * hd
    q = torch.randn(B * Lq, d, device=dev).bfloat16(); k = torch.randn(B * Lk, d, device=dev).bfloat16(); v = torch.randn(B * Lk, d, device=dev).bfloat16()
    o, lse = _ops.attn_fwd(q, k, v, B, H, Lq, Lk, hd, None, pd, 1234, 5, 0)
    do, dq, dk, dv = torch.randn_like(o), torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    mf = t_us(lambda: _ops.attn_fwd(q, k, v, B, H, Lq, Lk, hd, None, pd, 1234, 5, 0))
    mb = t_us(lambda: _ops.attn_bwd(q, k, v, o, do, dq, dk, dv, lse, B, H, Lq, Lk, hd, None, pd, 1234, 5, 0))
    fl = 4.0 * B * H * Lq * Lk * hd
    line = f"{name:20s} {Lq:4d} {Lk:4d} | {mf:6.1f} / {mb:6.1f} ({3 * fl / (mf + mb) / 1e6:5.0f})"
    try:
        q4 = q.view(B, Lq, H, hd).transpose(1, 2).contiguous().requires_grad_(True)
        k4 = k.view(B, Lk, H, hd).transpose(1, 2).contiguous().requires_grad_(True)
        v4 = v.view(B, Lk, H, hd).transpose(1, 2).contiguous().requires_grad_(True)
        vf = t_us(lambda: F.scaled_dot_product_attention(q4, k4, v4, dropout_p=pd))
        out = F.scaled_dot_product_attention(q4, k4, v4, dropout_p=pd)
        g = torch.randn_like(out)
        vb = t_us(lambda: torch.autograd.grad(out, (q4, k4, v4), g, retain_graph=True))
        line += f"   {vf:6.1f} / {vb:6.1f} ({3 * fl / (vf + vb) / 1e6:5.0f})"
    except Exception as ex:                                   # no fused backend for this shape on this build
        line += f"   vendor path unavailable: {type(ex).__name__}: {str(ex)[:80]}"
    print(line, flush=True)
