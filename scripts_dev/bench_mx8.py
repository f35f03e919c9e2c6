"""MX-fp8 GEMM vs the bf16 GEMM on the projection / FFN shapes of cfg 5 (d=1024, B*T_a = 12800 / 25600 rows) and cfg 2
(tuning aid): time incl. / excl. the activation quantiser, TFLOP/s against the 5 PF (fp8) and 2.5 PF (bf16) peaks."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd
from hri_emo_amd import _ops, _lib
L = _lib.lib()
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
shapes = [(12800, 3072, 1024), (12800, 1024, 1024), (12800, 4096, 1024), (12800, 1024, 4096), (12800, 2048, 1024), (4096, 3072, 1024), (4096, 1024, 4096),
          (25600, 3072, 768), (25600, 768, 768), (25600, 768, 3072), (25600, 2304, 768), (224, 1024, 1024), (224, 2048, 1024)]
for (M, N, K) in shapes:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5); b = torch.randn(N, device="cuda")
    w16 = w.bfloat16()
    t16 = timeit(lambda: _ops.linear_fwd(x, w16, b))
    xq, xs = _ops.quant_mx8(x); wq, ws = _ops.quant_mx8(w)
    tq = timeit(lambda: _ops.quant_mx8(x))
    row = f"M={M:6d} N={N:5d} K={K:5d}  bf16 {t16:7.1f} us {2*M*N*K/t16/1e6:6.0f} TF | quant {tq:6.1f} us ({3*M*K/tq/1e3:5.0f} GB/s)"
    for cfg in (0, 1, 2):
        L.hriemo_gemm_mx8_force_config(cfg)
        t8 = timeit(lambda: _ops.linear_fwd_mx8(xq, xs, wq, ws, b))
        row += f" | mx8 cfg{cfg} {t8:7.1f} us {2*M*N*K/t8/1e6:6.0f} TF"
    L.hriemo_gemm_mx8_force_config(-1)
    print(row, flush=True)
