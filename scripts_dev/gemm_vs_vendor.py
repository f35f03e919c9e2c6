"""Yardstick only (never on the product path): the hand-written GEMM next to torch.matmul (hipBLASLt / rocBLAS) on the
shapes of one cfg-2 step.  Same operands, bf16 in, events around 20 back-to-back launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd
from hri_emo_amd import _ops
dev = "cuda"

def t_us(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

nt = [(25600, 3072, 768), (25600, 2304, 768), (25600, 768, 768), (25600, 768, 3072), (25600, 1536, 768), (8192, 768, 768), (8192, 3072, 768), (384, 768, 768)]
nn = [(25600, 768, 3072), (25600, 3072, 768), (25600, 768, 768), (25600, 768, 2304), (8192, 768, 768)]
tn = [(3072, 768, 25600), (768, 3072, 25600), (768, 768, 25600), (2304, 768, 25600), (768, 768, 8192), (768, 768, 384)]
print("layout      M     N     K |  mine us (TF)   vendor us (TF)")
for (M, N, K) in nt:
    x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16(); b = torch.randn(N, device=dev)
    b16 = b.bfloat16()
    a = t_us(lambda: _ops.linear_fwd(x, w, b)); v = t_us(lambda: torch.nn.functional.linear(x, w, b16))
    fl = 2.0 * M * N * K
    print(f"NT     {M:6d} {N:5d} {K:5d} | {a:7.1f} ({fl/a/1e6:5.0f})  {v:7.1f} ({fl/v/1e6:5.0f})", flush=True)
for (M, Ko, Nr) in nn:
    dy = torch.randn(M, Nr, device=dev).bfloat16(); w = torch.randn(Nr, Ko, device=dev).bfloat16()
    a = t_us(lambda: _ops.linear_dx(dy, w)); v = t_us(lambda: dy @ w)
    fl = 2.0 * M * Ko * Nr
    print(f"NN     {M:6d} {Ko:5d} {Nr:5d} | {a:7.1f} ({fl/a/1e6:5.0f})  {v:7.1f} ({fl/v/1e6:5.0f})", flush=True)
for (No, Ko, Mr) in tn:
    dy = torch.randn(Mr, No, device=dev).bfloat16(); x = torch.randn(Mr, Ko, device=dev).bfloat16()
    out = torch.zeros((No, Ko), dtype=torch.float32, device=dev)
    a = t_us(lambda: _ops.linear_dw(dy, x, out, accumulate=True)); v = t_us(lambda: dy.t() @ x)     # vendor: bf16 out, no accumulate
    fl = 2.0 * No * Ko * Mr
    print(f"TN     {No:6d} {Ko:5d} {Mr:5d} | {a:7.1f} ({fl/a/1e6:5.0f})  {v:7.1f} ({fl/v/1e6:5.0f})", flush=True)
