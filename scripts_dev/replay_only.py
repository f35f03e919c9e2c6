"""The cfg-2 step captured once and replayed N times, nothing else (target for rocprofv3 timeline runs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd as H
from hri_emo_amd.dp import DataParallelStep
from hri_emo_amd.train import fusion_step_loss
from hri_emo_amd.optim import FusedClipAdamW
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.manual_seed(1234)
model = H.FusionWithEmotionDecoder(**bench.CFG).to(dev).train()
dp = DataParallelStep(model, fusion_step_loss, overlap=False)
opt = FusedClipAdamW(dp.buckets, lr=1e-4, weight_decay=1e-2, max_norm=5.0)
B = 64
dp.set_global_batch(B)
batch = bench.synth(B, 0, dev)
if os.environ.get("RAGGED", "0") == "1":      # bench.py's ragged leg: valid length ~U[L/2, L] per sample and modality
    g = torch.Generator().manual_seed(4321)
    la = torch.randint(bench.T_A // 2, bench.T_A + 1, (B,), generator=g)
    lt = torch.randint(bench.T_T // 2, bench.T_T + 1, (B,), generator=g)
    batch = (batch[0], batch[1], (torch.arange(bench.T_A)[None] >= la[:, None]).to(dev), (torch.arange(bench.T_T)[None] >= lt[:, None]).to(dev), batch[4])
if os.environ.get("VARLEN", "0") == "1":
    H.set_varlen(True)
dp.step(*batch)                   # eager once: sizes workspaces
if os.environ.get("EAGER", "0") != "1":
    dp.capture(*batch)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(5):
    dp.step(*batch)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    dp.step(*batch)
e1.record()
torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / n:.3f} ms/step")
