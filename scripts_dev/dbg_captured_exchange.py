"""captured gradient exchange at world size 1 (RCCL): which ingredient breaks the capture?  args: dropout snapshots(0/1) bucket_kb"""
import os, sys, socket, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import hri_emo_amd as H
from hri_emo_amd.dp import DataParallelStep
from hri_emo_amd.train import fusion_step_loss
p, snap, kb = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
with socket.socket() as so:
    so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.manual_seed(23)
m = H.FusionWithEmotionDecoder(d_model=256, num_emotions=5, n_heads=8, dropout=p).cuda().train()
g = torch.Generator().manual_seed(51)
B, Ta, Tt, d = 6, 90, 36, 256
h_a, h_t = torch.randn(B, Ta, d, generator=g).cuda().bfloat16(), torch.randn(B, Tt, d, generator=g).cuda().bfloat16()
la = torch.randint(Ta // 2, Ta + 1, (B,), generator=g); lt = torch.randint(Tt // 2, Tt + 1, (B,), generator=g)
m_a, m_t = (torch.arange(Ta)[None] >= la[:, None]).cuda(), (torch.arange(Tt)[None] >= lt[:, None]).cuda()
y = (torch.rand(B, 5, generator=g) < 0.3).float().cuda()
dp = DataParallelStep(m, fusion_step_loss, bucket_bytes=kb << 10, overlap=True, force_exchange=True)
print("buckets", len(dp.buckets.buckets), flush=True)
if snap:
    dp.buckets.enable_launch_snapshots()
dp.step(h_a, h_t, m_a, m_t, y)
torch.cuda.synchronize()
print("eager ok", flush=True)
dp.capture(h_a, h_t, m_a, m_t, y, collectives=True)
print("capture ok", flush=True)
for _ in range(3):
    dp.step(h_a, h_t, m_a, m_t, y)
torch.cuda.synchronize()
print("replays ok", dp.buckets.launch_snapshot_mismatches() if snap else "", flush=True)
dp.buckets.close()
dist.destroy_process_group()
