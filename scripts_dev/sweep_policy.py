"""Configurations 1 / 2 / 9 and the built-in choice (-1) on the projection shapes of cfg 4 (B=32, T_a=1000, T_t=50, d=768) and
cfg 5 (B=32, T_a=400, T_t=128, d=1024): does pick_config take the fastest one?  NT and NN, us per launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd  # noqa: F401
from hri_emo_amd import _ops, _lib
L = _lib.lib()
dev = torch.device("cuda", 0)


def timeit(fn, reps=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


shapes = [(32000, 768, 768), (32000, 2304, 768), (32000, 3072, 768), (32000, 768, 3072), (1600, 768, 768), (1600, 2304, 768), (1600, 3072, 768),
          (12800, 1024, 1024), (12800, 3072, 1024), (12800, 4096, 1024), (12800, 1024, 4096), (12800, 2048, 1024),
          (4096, 1024, 1024), (4096, 3072, 1024), (4096, 4096, 1024), (4096, 1024, 4096), (8192, 1536, 768)]
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16(); b = torch.randn(N, device=dev)
    dY = torch.randn(M, K, device=dev).bfloat16(); Wn = torch.randn(K, N, device=dev).bfloat16()
    row = f"{M:6d} x {N:5d} x {K:5d} |"
    for lay, fn in (("NT", lambda: _ops.linear_fwd(A, W, b)), ("NN", lambda: _ops.linear_dx(dY, Wn))):
        t = {}
        for cfg in (1, 2, 9, -1):
            L.hriemo_gemm_force_config(cfg)
            t[cfg] = timeit(fn)
        best = min((1, 2, 9), key=lambda c: t[c])
        row += f" {lay}: cfg1 {t[1]:6.1f} cfg2 {t[2]:6.1f} cfg9 {t[9]:6.1f} picked {t[-1]:6.1f} (best cfg{best}{'' if t[-1] <= 1.04 * t[best] else '  <-- MISS'}) |"
    print(row, flush=True)
L.hriemo_gemm_force_config(-1)
