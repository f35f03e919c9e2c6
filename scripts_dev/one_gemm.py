"""Run one GEMM shape/config a few times (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd
from hri_emo_amd import _ops, _lib
cfg, M, N, K = (int(x) for x in sys.argv[1:5])
layout = sys.argv[5] if len(sys.argv) > 5 else "NT"
L = _lib.lib(); L.hriemo_gemm_force_config(cfg)
torch.manual_seed(0)
if layout == "NT":
    A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda")
    fn = lambda: _ops.linear_fwd(A, W, b)
elif layout == "NN":
    A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(K, N, device="cuda").bfloat16()
    fn = lambda: _ops.linear_dx(A, W)
else:
    dY = torch.randn(K, M, device="cuda").bfloat16(); X = torch.randn(K, N, device="cuda").bfloat16(); out = torch.empty(M, N, device="cuda")
    fn = lambda: _ops.linear_dw(dY, X, out)
for _ in range(6): fn()
torch.cuda.synchronize()
