"""Debug aid: one rank over RCCL with force_exchange (hooks on, all-reduce = identity): hooks-on gradients vs hooks-off gradients"""
import os, sys, socket, torch, torch.distributed as dist
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
with socket.socket() as so:
    so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import hri_emo_amd as H
from hri_emo_amd import _ops
from hri_emo_amd.dp import DataParallelStep
from hri_emo_amd.train import fusion_step_loss
g = torch.Generator().manual_seed(31)
B, Ta, Tt, d = 4, 48, 24, 128
h_a, h_t = torch.randn(B, Ta, d, generator=g), torch.randn(B, Tt, d, generator=g)
m_a = torch.arange(Ta)[None] >= torch.randint(Ta // 2, Ta + 1, (B, 1), generator=g)
m_t = torch.arange(Tt)[None] >= torch.randint(Tt // 2, Tt + 1, (B, 1), generator=g)
y = (torch.rand(B, 4, generator=torch.Generator().manual_seed(5)) < 0.3).float()
torch.manual_seed(3)
m = H.FusionWithEmotionDecoder(d_model=d, num_emotions=4, n_heads=8, dropout=0.0).cuda().train()
dp = DataParallelStep(m, fusion_step_loss, bucket_bytes=256 << 10, overlap=True, force_exchange=True)
batch = (h_a.cuda().bfloat16(), h_t.cuda().bfloat16(), m_a.cuda(), m_t.cuda(), y.cuda())
names = {id(p): n for n, p in m.named_parameters()}
log = []
orig_launch = dp.buckets._launch
def traced(bi):
    log.append(("launch", bi, "side" if torch.cuda.current_stream() == _ops._side_streams.get(0) else "main"))
    return orig_launch(bi)
dp.buckets._launch = traced
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    dp.buckets.suspended = True
    dp._fwd_bwd(*batch); torch.cuda.synchronize()
    want = dp.buckets.flat.clone()
    dp.buckets.suspended = False
    del log[:]
    dp.step(*batch); torch.cuda.synchronize()
    got = dp.buckets.flat
    bad = []
    for p in dp.buckets.params:
        o, n = dp.buckets._offsets[id(p)], p.numel()
        r = float((got[o:o + n] - want[o:o + n]).norm() / want[o:o + n].norm().clamp_min(1e-20))
        if r > 1e-5:
            bad.append((dp.buckets._bucket_of[id(p)], names[id(p)], round(r, 4)))
    print(f"rep {rep}: {len(bad)} bad params {bad[:12]}")
    print("   launches:", log)
dist.destroy_process_group()
