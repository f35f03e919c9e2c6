"""GEMM tile-configuration sweep on the real shapes of BASELINE cfg 2 (tuning aid, not a test)."""
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
    for (M, N, K) in [(200, 136, 96), (300, 264, 160), (512, 512, 256)]:
        A, W, b = ints((M, K), 1), ints((N, K), 2), ints((N,), 3)
        y = _ops.linear_fwd(A.to(dev).bfloat16(), W.to(dev).bfloat16(), b.to(dev))
        ok1 = torch.equal(y.float().cpu(), (A @ W.t() + b).bfloat16().float())
        dY, W2 = ints((M, N), 4), ints((N, K), 5)
        dx = _ops.linear_dx(dY.to(dev).bfloat16(), W2.to(dev).bfloat16())
        ok2 = torch.equal(dx.float().cpu(), (dY @ W2).bfloat16().float())
        X = ints((M, K), 6)
        out = torch.empty((N, K), dtype=torch.float32, device=dev)
        _ops.linear_dw(dY.to(dev).bfloat16(), X.to(dev).bfloat16(), out)
        ok3 = torch.equal(out.cpu(), dY.t() @ X)
        if not (ok1 and ok2 and ok3): return False, (M, N, K, ok1, ok2, ok3)
    return True, None
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3   # us
shapes_nt = [(384, 768, 768), (384, 2304, 768), (384, 2048, 768), (384, 768, 2048), (64, 256, 3072), (25600, 3072, 768), (25600, 2304, 768), (25600, 768, 768), (25600, 768, 3072), (25600, 1536, 768), (8192, 768, 768), (8192, 3072, 768), (8192, 2304, 768)]
shapes_nn = [(384, 768, 2304), (384, 2048, 768), (384, 768, 2048), (25600, 768, 3072), (25600, 3072, 768), (25600, 768, 768), (25600, 768, 2304), (8192, 768, 768), (8192, 768, 3072)]   # (M, Kout, Nred)
shapes_tn = [(3072, 768, 25600), (768, 3072, 25600), (768, 768, 25600), (2304, 768, 25600), (768, 768, 8192), (3072, 768, 8192)]      # (Nout, Kout, Mred)
torch.manual_seed(0)
res = {}
ncfg = int(os.environ.get("NCFG", "6"))
cfg_list = [int(x) for x in os.environ["CFGS"].split(",")] if "CFGS" in os.environ else list(range(ncfg))
for cfg in cfg_list:
    ok, info = check(cfg)
    print(f"cfg {cfg}: exact={ok} {info or ''}", flush=True)
    if not ok: continue
    L.hriemo_gemm_force_config(cfg)
    for (M, N, K) in shapes_nt:
        A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16(); b = torch.randn(N, device=dev)
        us = timeit(lambda: _ops.linear_fwd(A, W, b))
        res[("NT", M, N, K, cfg)] = us
    for (M, Ko, Nr) in shapes_nn:
        dY = torch.randn(M, Nr, device=dev).bfloat16(); W = torch.randn(Nr, Ko, device=dev).bfloat16()
        us = timeit(lambda: _ops.linear_dx(dY, W))
        res[("NN", M, Ko, Nr, cfg)] = us
    for (No, Ko, Mr) in shapes_tn:
        dY = torch.randn(Mr, No, device=dev).bfloat16(); X = torch.randn(Mr, Ko, device=dev).bfloat16()
        out = torch.empty((No, Ko), dtype=torch.float32, device=dev)
        us = timeit(lambda: _ops.linear_dw(dY, X, out))
        res[("TN", No, Ko, Mr, cfg)] = us
keys = sorted(set(k[:4] for k in res))
print("layout  M      N      K     | " + " ".join(f"cfg{c}:us/TF   " for c in range(ncfg)))
for k in keys:
    fl = 2.0 * k[1] * k[2] * k[3]
    row = []
    for c in cfg_list:
        us = res.get(k + (c,))
        row.append(f"{us:7.1f}/{fl/us/1e6:5.0f}" if us else "      -      ")
    print(f"{k[0]:3s} {k[1]:6d} {k[2]:6d} {k[3]:6d} | " + " ".join(row))
