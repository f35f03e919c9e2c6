"""Linear + LayerNorm in one kernel (hriemo_gemm_ln_fwd) against the two launches it replaces, alone on the chip, at the
fusion layers' shapes.  usage: python scripts_dev/bench_gemm_ln.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd  # noqa: F401,E402
from hri_emo_amd import _ops  # noqa: E402


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


g = torch.Generator().manual_seed(0)
for M, d, K in ((25600, 768, 768), (25600, 768, 3072), (8192, 768, 768), (8192, 768, 3072), (18432, 768, 768)):
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(d, K, generator=g) / K ** 0.5).bfloat16().cuda()
    b = torch.randn(d, generator=g).cuda()
    X32 = torch.randn(M, d, generator=g).cuda()
    X16 = X32.bfloat16()
    gamma, beta = torch.ones(d).cuda(), torch.zeros(d).cuda()
    for p in (0.0, 0.1):
        t_gemm = timeit(lambda: _ops.linear_fwd(A, W, b))
        gref = _ops.linear_fwd(A, W, b)
        t_ln = timeit(lambda: _ops.add_ln_fwd(gref, X16, gamma, beta, p, 1, 2, 0, x32=X32, want32=True))
        t_both = timeit(lambda: _ops.add_ln_fwd(_ops.linear_fwd(A, W, b), X16, gamma, beta, p, 1, 2, 0, x32=X32, want32=True))
        t_fused = timeit(lambda: _ops.proj_add_ln_fwd(A, W, b, X16, X32, gamma, beta, p, 1, 2, 0, True))
        tf = 2.0 * M * d * K / 1e6
        print(f"M={M} d={d} K={K} p={p}: gemm {t_gemm:6.1f} us ({tf / t_gemm:5.0f} TF/s) + add_ln {t_ln:5.1f} us = {t_gemm + t_ln:6.1f} "
              f"(back to back {t_both:6.1f}) | fused {t_fused:6.1f} us ({tf / t_fused:5.0f} TF/s)  ratio {t_fused / t_both:.2f}", flush=True)
