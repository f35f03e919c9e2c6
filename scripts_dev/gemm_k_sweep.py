"""Fixed cost per launch of the GEMM kernel: time against K at fixed M x N (operands aliased to row 0 and real), so that
t(K) = a + b*K separates the per-output cost (epilogue, tile boundaries, launch) from the main loop."""
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
Ks = [256, 512, 768, 1536, 3072, 6144]
for (M, N, cfg) in [(25600, 768, 1), (25600, 768, 2), (25600, 3072, 2), (25600, 3072, 1)]:
    L.hriemo_gemm_force_config(cfg)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    rows = []
    for K in Ks:
        A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16()
        tr = timeit(lambda: _ops.gemm(0, 0, M, N, K, A, K, W, K, y, N))
        ta = timeit(lambda: _ops.gemm(0, 0, M, N, K, A, 0, W, 0, y, N))
        rows.append((K, tr, ta))
    b_r = (rows[-1][1] - rows[2][1]) / (Ks[-1] - Ks[2]); a_r = rows[2][1] - b_r * Ks[2]
    b_a = (rows[-1][2] - rows[2][2]) / (Ks[-1] - Ks[2]); a_a = rows[2][2] - b_a * Ks[2]
    peak_b = 2.0 * M * N / 2.5e9          # us per unit of K at the dense bf16 peak
    print(f"NT M={M} N={N} cfg {cfg}: " + "  ".join(f"K={k}: {tr:.1f}/{ta:.1f}" for k, tr, ta in rows))
    print(f"    real: a = {a_r:.1f} us, b = {b_r * 1e3:.2f} ns per k ({peak_b / b_r * 100:.0f} % of peak in the loop);  aliased: a = {a_a:.1f} us, b = {b_a * 1e3:.2f} ns per k ({peak_b / b_a * 100:.0f} %)")
L.hriemo_gemm_force_config(-1)
# the same with the OUTPUT aliased as well (ldc = 0: every tile row stores to row 0): what is left of `a` is not store bandwidth
for (M, N, cfg) in [(25600, 768, 1), (25600, 3072, 2)]:
    L.hriemo_gemm_force_config(cfg)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out = []
    for K in (256, 768, 1536):
        A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16()
        out.append((K, timeit(lambda: _ops.gemm(0, 0, M, N, K, A, 0, W, 0, y, N)), timeit(lambda: _ops.gemm(0, 0, M, N, K, A, 0, W, 0, y, 0))))
    print(f"NT M={M} N={N} cfg {cfg}, operands aliased, output real / aliased: " + "  ".join(f"K={k}: {t1:.1f}/{t2:.1f}" for k, t1, t2 in out))
L.hriemo_gemm_force_config(-1)
