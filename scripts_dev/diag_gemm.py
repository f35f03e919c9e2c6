import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd
from hri_emo_amd import _ops, _lib
lib = os.environ.get("HRIEMO_LIB")
if lib: _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), lib)
L = _lib.lib()
def ints(shape, lo=-3, hi=4, seed=0):
    g = torch.Generator().manual_seed(seed); return torch.randint(lo, hi, shape, generator=g).float()
M, N, K = 1024, 768, 768
A, W, b = ints((M, K), seed=1), ints((N, K), seed=2), ints((N,), seed=3)
ref = A @ W.t() + b
for cfg in (-1, 0, 1, 2, 3):
    L.hriemo_gemm_force_config(cfg)
    for relu in (False, True):
        for rep in range(2):
            y = _ops.linear_fwd(A.cuda().bfloat16(), W.cuda().bfloat16(), b.cuda(), relu=relu).float().cpu()
            r = (ref.clamp(min=0) if relu else ref).bfloat16().float()
            bad = (y != r).nonzero()
            print(f"cfg {cfg} relu {relu} rep {rep}: bad {len(bad)}", "rows", sorted(set((bad[:, 0] // 16).tolist()))[:12], "cols", sorted(set((bad[:, 1] // 16).tolist()))[:12], flush=True)
            if len(bad):
                i, j = bad[0].tolist(); print("   first", i, j, float(y[i, j]), float(r[i, j]), float(ref[i, j]))
