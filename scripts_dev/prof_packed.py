"""The headline step on the bench's ragged batch (valid fraction ~0.72), captured and replayed: `padded` (masks only) or `packed`
(varlen encoder).  Run each under rocprofv3 --kernel-trace --stats and compare the per-kernel totals: which kernels do not
scale with the valid rows.  usage: python scripts_dev/prof_packed.py padded|packed [replays]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import hri_emo_amd as H  # noqa: E402
from hri_emo_amd.dp import DataParallelStep  # noqa: E402
from hri_emo_amd.train import fusion_step_loss  # noqa: E402

mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = H.FusionWithEmotionDecoder(**bench.CFG).to(dev).train()
dp = DataParallelStep(model, fusion_step_loss, overlap=False)
B, T_A, T_T = 64, bench.T_A, bench.T_T
dp.set_global_batch(B)
batch = bench.synth(B, 0, dev)
g = torch.Generator().manual_seed(4321)
la = torch.randint(T_A // 2, T_A + 1, (B,), generator=g)
lt = torch.randint(T_T // 2, T_T + 1, (B,), generator=g)
rb = (batch[0], batch[1], (torch.arange(T_A)[None] >= la[:, None]).to(dev), (torch.arange(T_T)[None] >= lt[:, None]).to(dev), batch[4])
if mode == "packed":
    H.set_varlen(True)
dp.step(*rb)
dp.capture(*rb)
for _ in range(3):
    dp.step(*rb)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n):
    dp.step(*rb)
torch.cuda.synchronize()
print(f"{mode}: {(time.perf_counter() - t) / n * 1e3:.3f} ms/step over {n} replays, valid fraction "
      f"{float((la.sum() / T_A + lt.sum() / T_T) / (2 * B)):.3f}")
