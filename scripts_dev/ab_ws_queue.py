"""Configuration 9 alone: work queue (debug flags bit 3 clear) against the static walk (bit 3 set), interleaved rounds in one
process, on the shapes of the cfg-2 step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd  # noqa: F401
from hri_emo_amd import _ops, _lib
L = _lib.lib()
dev = torch.device("cuda", 0)


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


L.hriemo_gemm_force_config(9)
for lay, M, N, K in [("NT", 25600, 768, 768), ("NT", 25600, 768, 3072), ("NT", 25600, 1536, 768), ("NN", 25600, 768, 768), ("NN", 25600, 768, 2304),
                     ("NN", 25600, 768, 3072), ("TN", 768, 768, 25600), ("TN", 3072, 768, 25600)]:
    if lay == "NT":
        A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16(); b = torch.randn(N, device=dev)
        fn = lambda: _ops.linear_fwd(A, W, b)
    elif lay == "NN":
        dY = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(K, N, device=dev).bfloat16()
        fn = lambda: _ops.linear_dx(dY, W)
    else:
        dY = torch.randn(K, M, device=dev).bfloat16(); X = torch.randn(K, N, device=dev).bfloat16(); out = torch.zeros(M, N, device=dev)
        fn = lambda: _ops.linear_dw(dY, X, out)
    r = {1: [], 9: []}
    for rnd in range(5):
        for f in (1, 9):
            L.hriemo_gemm_debug_flags(f)
            r[f].append(timeit(fn))
    q, st = sorted(r[1])[2], sorted(r[9])[2]
    print(f"{lay} {M}x{N}x{K}: queue {q:6.1f} us   static {st:6.1f} us   queue / static {q / st:.3f}", flush=True)
L.hriemo_gemm_force_config(-1)
L.hriemo_gemm_debug_flags(9)
