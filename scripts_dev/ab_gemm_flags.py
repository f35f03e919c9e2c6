"""A/B of GemmArgs.flags bit 0 (the first K-step after an epilogue counts the epilogue's stores in its retire wait) in ONE process:
(1) isolated cfg-1 (256x128 x 3 stages) launches on the cfg-2 shapes, interleaved rounds; exactness on integer operands;
(2) the captured cfg-2 step, one capture per setting, replays interleaved."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd as H
from hri_emo_amd import _ops, _lib
from hri_emo_amd.dp import DataParallelStep
from hri_emo_amd.train import fusion_step_loss
import bench
dev = torch.device("cuda", 0)
L = _lib.lib()


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def ints(shape, seed):
    g = torch.Generator().manual_seed(seed); return torch.randint(-3, 4, shape, generator=g).float()


# exactness with the flag on (persistent multi-tile walk, edge tiles, every epilogue)
L.hriemo_gemm_debug_flags(1)
L.hriemo_gemm_force_config(int(os.environ.get('CFG', '9')))
for (M, N, K) in [(2000, 392, 448), (25600, 768, 768), (5000, 768, 1024), (3000, 136, 768), (70000, 768, 128), (66000, 264, 192)]:
    A, W, b, R = ints((M, K), 1), ints((N, K), 2), ints((N,), 3), ints((M, N), 4)
    Ad, Wd = A.to(dev).bfloat16(), W.to(dev).bfloat16()
    y = _ops.linear_fwd(Ad, Wd, b.to(dev))
    ok = torch.equal(y.float().cpu(), (A @ W.t() + b).bfloat16().float())
    y1 = _ops.linear_fwd(Ad, Wd, b.to(dev), relu=True)
    ok &= torch.equal(y1.float().cpu(), torch.relu(A @ W.t() + b).bfloat16().float())
    dY, W2 = ints((M, N), 5), ints((N, K), 6)
    dx = _ops.linear_dx(dY.to(dev).bfloat16(), W2.to(dev).bfloat16(), epi=3, aux=ints((M, K), 7).to(dev).bfloat16())
    ok &= torch.equal(dx.float().cpu(), (dY @ W2 + ints((M, K), 7)).bfloat16().float())
    print(f"exact {M}x{N}x{K}: {ok}", flush=True)
L.hriemo_gemm_force_config(-1)
if os.environ.get('FULL', '0') != '1':
    sys.exit(0)

shapes = [("NT", 25600, 768, 768), ("NT", 25600, 768, 3072), ("NT", 8192, 768, 768), ("NT", 25600, 1536, 768), ("NN", 25600, 768, 768), ("NN", 25600, 768, 2304),
          ("NN", 25600, 768, 3072), ("NN", 8192, 768, 768)]
for lay, M, N, K in shapes:
    if lay == "NT":
        A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16(); b = torch.randn(N, device=dev)
        fn = lambda: _ops.linear_fwd(A, W, b)
    else:
        dY = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(K, N, device=dev).bfloat16()
        fn = lambda: _ops.linear_dx(dY, W)
    r = {0: [], 1: []}
    for rnd in range(5):
        for f in (0, 1):
            L.hriemo_gemm_debug_flags(f)
            r[f].append(timeit(fn))
    m0, m1 = sorted(r[0])[2], sorted(r[1])[2]
    print(f"{lay} {M}x{N}x{K}: flags 0 median {m0:6.1f} us (min {min(r[0]):6.1f})   flags 1 median {m1:6.1f} us (min {min(r[1]):6.1f})   {100 * (m1 / m0 - 1):+.1f} %", flush=True)

# the captured step
torch.manual_seed(1234)
steps = {}
for f in (0, 1):
    L.hriemo_gemm_debug_flags(f)
    torch.manual_seed(1234)
    model = H.FusionWithEmotionDecoder(**bench.CFG).to(dev).train()
    dp = DataParallelStep(model, fusion_step_loss, overlap=False)
    dp.set_global_batch(64)
    batch = bench.synth(64, 0, dev)
    dp.step(*batch)
    dp.capture(*batch)
    steps[f] = (dp, batch)
res = {0: [], 1: []}
for rnd in range(5):
    for f in (0, 1):
        dp, batch = steps[f]
        for _ in range(3): dp.step(*batch)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): dp.step(*batch)
        e1.record(); torch.cuda.synchronize()
        res[f].append(e0.elapsed_time(e1) / 30)
print("captured cfg-2 step: flags 0", " ".join(f"{x:.3f}" for x in res[0]), "| flags 1", " ".join(f"{x:.3f}" for x in res[1]), flush=True)
L.hriemo_gemm_debug_flags(9)
