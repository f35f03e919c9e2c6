"""Same-process A/B of hriemo_gemm_debug_flags values on the captured cfg-2 step: one capture per value (the tile
configuration and the kernels' flag word are baked into the graph), replays interleaved.  usage: ab_step_flags.py 1 5 9 ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd as H
from hri_emo_amd import _lib
from hri_emo_amd.dp import DataParallelStep
from hri_emo_amd.train import fusion_step_loss
import bench
dev = torch.device("cuda", 0)
L = _lib.lib()
vals = [int(x) for x in sys.argv[1:]] or [1, 3]
steps = {}
for f in vals:
    L.hriemo_gemm_debug_flags(f)
    torch.manual_seed(1234)
    model = H.FusionWithEmotionDecoder(**bench.CFG).to(dev).train()
    dp = DataParallelStep(model, fusion_step_loss, overlap=False)
    dp.set_global_batch(64)
    batch = bench.synth(64, 0, dev)
    dp.step(*batch)
    dp.capture(*batch)
    steps[f] = (dp, batch)
res = {f: [] for f in vals}
for rnd in range(5):
    for f in vals:
        dp, batch = steps[f]
        for _ in range(3): dp.step(*batch)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): dp.step(*batch)
        e1.record(); torch.cuda.synchronize()
        res[f].append(e0.elapsed_time(e1) / 30)
for f in vals:
    r = sorted(res[f])
    print(f"flags {f:3d}: median {r[2]:.3f} ms  (" + " ".join(f"{x:.3f}" for x in res[f]) + ")", flush=True)
L.hriemo_gemm_debug_flags(9)
