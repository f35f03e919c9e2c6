"""GEMM time while another kernel holds some CUs (stand-in for an RCCL collective): the work-queue kernels (configs 1 / 2) against
the loader / consumer kernel (config 9, static walk)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd
from hri_emo_amd import _ops, _lib
L = _lib.lib()
sink = torch.zeros(4, device="cuda")
side = torch.cuda.Stream()
def timed(fn, hog_blocks):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    if hog_blocks:
        L.hriemo_debug_hog(hog_blocks, 6000, sink.data_ptr(), side.cuda_stream)
        torch.cuda._sleep(2_000_000)          # let the hog settle on its CUs first
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 10 * 1e3
for (M, N, K) in [(25600, 768, 768), (25600, 3072, 768), (25600, 768, 3072)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda")
    f = lambda: _ops.linear_fwd(A, W, b)
    for cfg in (1, 2, 9):
        L.hriemo_gemm_force_config(cfg)
        print(f"NT {M}x{N}x{K} cfg {cfg}: alone {timed(f, 0):7.1f} us   with 32 CUs held {timed(f, 32):7.1f} us   with 64 held {timed(f, 64):7.1f} us", flush=True)
    L.hriemo_gemm_force_config(-1)
