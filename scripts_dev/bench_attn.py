"""Attention core timings on the cfg-2 shapes (tuning aid): TFLOP/s (algorithmic: fwd 4*B*H*Lq*Lk*hd, bwd 2x) and
fraction of the attention roofline min(MFMA peak, AI * HBM), AI = Lq*Lk/(Lq+Lk) flop/B."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd
from hri_emo_amd import _ops
B, H, hd = 64, 8, 96
d = H * hd
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for name, Lq, Lk in [("self_a", 400, 400), ("a2t (q=audio,k=text)", 400, 128), ("t2a (q=text,k=audio)", 128, 400), ("self_t", 128, 128), ("decoder cross", 6, 128)]:
    for p in (0.0, 0.1):
        q = torch.randn(B * Lq, d, device="cuda").bfloat16(); k = torch.randn(B * Lk, d, device="cuda").bfloat16(); v = torch.randn(B * Lk, d, device="cuda").bfloat16()
        o, lse, mb = _ops.attn_fwd(q, k, v, B, H, Lq, Lk, hd, None, p, 1234, 5, 0, want_bits=True)
        do = torch.randn_like(o); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
        tf = timeit(lambda: _ops.attn_fwd(q, k, v, B, H, Lq, Lk, hd, None, p, 1234, 5, 0, want_bits=True))
        tb = timeit(lambda: _ops.attn_bwd(q, k, v, o, do, dq, dk, dv, lse, B, H, Lq, Lk, hd, None, p, 1234, 5, 0, mask_bits=mb))
        fl = 4.0 * B * H * Lq * Lk * hd
        ai = Lq * Lk / (Lq + Lk)
        roof = min(2500.0, ai * 8.0)            # TFLOP/s
        print(f"{name:24s} p={p:3.1f}  fwd {tf:6.1f} us {fl/tf/1e6:6.0f} TF ({fl/tf/1e6/roof*100:4.1f}% of {roof:5.0f} roofline, {fl/tf/1e6/25:4.1f}% MFMA)   "
              f"bwd {tb:6.1f} us {2*fl/tb/1e6:6.0f} TF ({2*fl/tb/1e6/roof*100:4.1f}%)   fwd+bwd {3*fl/(tf+tb)/1e6:6.0f} TF", flush=True)
