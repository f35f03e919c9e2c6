"""Latency of the decoder / gate sized GEMMs (M = 384 / 64) per tile configuration: back-to-back launches on one stream
(each waits for the previous one, so the average is the kernel's own latency + the launch gap).  Tuning aid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd
from hri_emo_amd import _ops, _lib
L = _lib.lib()
dev = "cuda"
def ints(shape, seed):
    g = torch.Generator().manual_seed(seed); return torch.randint(-3, 4, shape, generator=g).float()
def check(cfg):
    L.hriemo_gemm_force_config(cfg)
    for (M, N, K) in [(200, 136, 512), (384, 264, 768), (70, 512, 1024)]:
        A, W, b = ints((M, K), 1), ints((N, K), 2), ints((N,), 3)
        y = _ops.linear_fwd(A.to(dev).bfloat16(), W.to(dev).bfloat16(), b.to(dev))
        ok1 = torch.equal(y.float().cpu(), (A @ W.t() + b).bfloat16().float())
        dY, W2 = ints((M, N), 4), ints((N, K), 5)
        dx = _ops.linear_dx(dY.to(dev).bfloat16(), W2.to(dev).bfloat16())
        ok2 = torch.equal(dx.float().cpu(), (dY @ W2).bfloat16().float())
        if not (ok1 and ok2): return False, (M, N, K, ok1, ok2)
    return True, None
def timeit(fn, reps=100):
    """reps launches captured into one graph and replayed: kernel latency + the gap between dependent graph nodes, as in the step"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (5 * reps) * 1e3   # us
nt = [(384, 768, 768), (384, 2304, 768), (384, 2048, 768), (384, 768, 2048), (384, 1536, 768), (64, 256, 3072), (64, 768, 256)]
nn = [(384, 768, 768), (384, 768, 2304), (384, 2048, 768), (384, 768, 2048), (64, 3072, 256)]      # (M, Kout, Nred)
cfgs = [int(x) for x in os.environ.get("CFGS", "3,6,7,0").split(",")]
res = {}
for cfg in cfgs:
    ok, info = check(cfg)
    print(f"cfg {cfg}: exact={ok} {info or ''}", flush=True)
    if not ok: continue
    L.hriemo_gemm_force_config(cfg)
    for (M, N, K) in nt:
        A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16(); b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        res[("NT", M, N, K, cfg)] = timeit(lambda: _ops.gemm(0, 0, M, N, K, A, K, W, K, y, N, bias=b))
    for (M, Ko, Nr) in nn:
        dY = torch.randn(M, Nr, device=dev).bfloat16(); W = torch.randn(Nr, Ko, device=dev).bfloat16()
        y = torch.empty(M, Ko, device=dev, dtype=torch.bfloat16)
        res[("NN", M, Ko, Nr, cfg)] = timeit(lambda: _ops.gemm(0, 1, M, Ko, Nr, dY, Nr, W, Ko, y, Ko))
L.hriemo_gemm_force_config(-1)
print("layout     M     N     K | " + " ".join(f"cfg{c:<2d} us" for c in cfgs))
for k in sorted(set(k[:4] for k in res)):
    print(f"{k[0]:6s} {k[1]:5d} {k[2]:5d} {k[3]:5d} | " + " ".join(f"{res.get(k + (c,), float('nan')):8.2f}" for c in cfgs))
