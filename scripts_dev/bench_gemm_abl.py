"""Timing-only ablation of the GEMM main loop (outputs are wrong by construction): which of LDS-DMA issue,
LDS fragment reads or the MFMA stream bounds the kernel.  HRIEMO_LIB selects the library build."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd
from hri_emo_amd import _lib, _ops
lib = os.environ.get("HRIEMO_LIB")
if lib:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), lib)
L = _lib.lib()
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
out = []
for cfg in (0, 1, 2, 4):
    L.hriemo_gemm_force_config(cfg)
    for (M, N, K) in [(25600, 3072, 768), (25600, 768, 3072), (25600, 768, 768), (25600, 2304, 768)]:
        A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda")
        us = timeit(lambda: _ops.linear_fwd(A, W, b))
        out.append(f"cfg{cfg} NT {M}x{N}x{K}: {us:7.1f} us {2.0*M*N*K/us/1e6:6.0f} TF")
print(lib or "libhriemo.so", " | ".join(out))
