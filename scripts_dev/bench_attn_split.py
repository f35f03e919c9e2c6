"""dQ and dK/dV kernel times per attention site (library profiler classes), narrow vs wide backward tiles
(HRIEMO_ATTN_WIDE_BWD).  Tuning aid."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd
from hri_emo_amd import _ops, _lib
L = _lib.lib()
B, H, hd, p = 64, 8, 96, 0.1
d = H * hd
def collect():
    out = {}
    for c in range(L.hriemo_prof_nclass()):
        ms, n, w = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
        L.hriemo_prof_collect(c, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(w))
        if n.value:
            out[L.hriemo_prof_name(c).decode()] = ms.value / n.value * 1e3
    return out
for name, Lq, Lk in [("self_a", 400, 400), ("a2t", 400, 128), ("t2a", 128, 400), ("self_t", 128, 128)]:
    q = torch.randn(B * Lq, d, device="cuda").bfloat16(); k = torch.randn(B * Lk, d, device="cuda").bfloat16(); v = torch.randn(B * Lk, d, device="cuda").bfloat16()
    o, lse = _ops.attn_fwd(q, k, v, B, H, Lq, Lk, hd, None, p, 1234, 5, 0)
    do = torch.randn_like(o); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    for _ in range(3): _ops.attn_bwd(q, k, v, o, do, dq, dk, dv, lse, B, H, Lq, Lk, hd, None, p, 1234, 5, 0)
    torch.cuda.synchronize()
    L.hriemo_prof_enable(1)
    for _ in range(10): _ops.attn_bwd(q, k, v, o, do, dq, dk, dv, lse, B, H, Lq, Lk, hd, None, p, 1234, 5, 0)
    torch.cuda.synchronize()
    r = collect()
    L.hriemo_prof_enable(0)
    print(f"wide_bwd={os.environ.get('HRIEMO_ATTN_WIDE_BWD', '0')} {name:7s} Lq={Lq} Lk={Lk}: " + "  ".join(f"{k_} {v_:6.1f} us" for k_, v_ in r.items() if "attn" in k_), flush=True)
