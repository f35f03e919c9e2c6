"""Where does the GEMM kernel's time go: the same launches with every operand row aliased to row 0 (lda = ldb = 0: all
LDS-DMA loads hit one cache line set) against the real strides.  The aliased run is the instruction schedule alone; the gap to
the real run is what memory (L2 -> LDS latency / bandwidth) costs.  Results of the aliased run are garbage by construction."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd
from hri_emo_amd import _ops, _lib
L = _lib.lib()
dev = "cuda"
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
print("layout cfg      M     N     K |   real us  TF/s | aliased us  TF/s")
for (M, N, K) in [(25600, 768, 768), (25600, 3072, 768), (25600, 768, 3072), (25600, 2304, 768)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16(); Wt = torch.randn(K, N, device=dev).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    for cfg in (1, 2, 9):
        L.hriemo_gemm_force_config(cfg)
        for lay, fn_real, fn_alias in (("NT", lambda: _ops.gemm(0, 0, M, N, K, A, K, W, K, y, N), lambda: _ops.gemm(0, 0, M, N, K, A, 0, W, 0, y, N)),
                                        ("NN", lambda: _ops.gemm(0, 1, M, N, K, A, K, Wt, N, y, N), lambda: _ops.gemm(0, 1, M, N, K, A, 0, Wt, 0, y, N))):
            tr, ta = timeit(fn_real), timeit(fn_alias)
            print(f"{lay:6s} {cfg:3d} {M:6d} {N:5d} {K:5d} | {tr:9.1f} {fl / tr / 1e6:5.0f} | {ta:10.1f} {fl / ta / 1e6:5.0f}")
L.hriemo_gemm_force_config(-1)
