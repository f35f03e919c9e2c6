"""Where the GEMM time of one cfg-2 step goes: every GEMM launch of a step timed in place (events around the launch,
single stream) next to the same launch repeated alone on the same operands (tuning aid, not a test)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd as H
from hri_emo_amd import _ops
from hri_emo_amd.dp import DataParallelStep
from hri_emo_amd.train import fusion_step_loss
import bench

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = H.FusionWithEmotionDecoder(**bench.CFG).to(dev).train()
dp = DataParallelStep(model, fusion_step_loss, overlap=False)
B = 64
dp.set_global_batch(B)
batch = bench.synth(B, 0, dev)
_ops.side_stream(dev)
_ops.TWO_STREAMS = False
for _ in range(3):
    dp._fwd_bwd(*batch)
torch.cuda.synchronize()

orig = _ops.gemm
log = []
def timed(ta, tb, M, N, K, A, lda, Bm, ldb, C, ldc, c_f32=False, bias=None, epi=0, aux=None, ldaux=0, accumulate=False):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(ta, tb, M, N, K, A, lda, Bm, ldb, C, ldc, c_f32, bias, epi, aux, ldaux, accumulate)
    e1.record()
    key = (ta, tb, M, N, K, int(c_f32), epi, int(bias is not None), int(accumulate))
    log.append((key, e0, e1, (ta, tb, M, N, K, A, lda, Bm, ldb, C, ldc, c_f32, bias, epi, aux, ldaux, accumulate)))
_ops.gemm = timed
NSTEP = 3
for _ in range(NSTEP):
    dp._fwd_bwd(*batch)
torch.cuda.synchronize()
_ops.gemm = orig

inst = collections.defaultdict(list)
args = {}
for key, e0, e1, a in log:
    inst[key].append(e0.elapsed_time(e1) * 1e3)
    args[key] = a

def alone(a, reps=10):
    # accumulate=True launches would keep adding into C: harmless for timing
    for _ in range(2):
        orig(*a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        orig(*a)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

rows = []
for key, ts in inst.items():
    n = len(ts) / NSTEP
    t_in = sum(ts) / len(ts)
    t_al = alone(args[key])
    fl = 2.0 * key[2] * key[3] * key[4]
    rows.append((n * t_in, key, n, t_in, t_al, fl))
rows.sort(reverse=True)
print("ta tb      M     N     K f32 epi bias acc |  n/step  in-step us (TF)   alone us (TF)   step share us")
tot_in = tot_al = tot_fl = 0.0
for share, key, n, t_in, t_al, fl in rows:
    print(f"{key[0]:2d} {key[1]:2d} {key[2]:6d} {key[3]:5d} {key[4]:5d} {key[5]:3d} {key[6]:3d} {key[7]:4d} {key[8]:3d} | {n:6.1f}  "
          f"{t_in:8.1f} ({fl / t_in / 1e6:5.0f})  {t_al:8.1f} ({fl / t_al / 1e6:5.0f})  {share:8.1f}")
    tot_in += n * t_in; tot_al += n * t_al; tot_fl += n * fl
print(f"GEMM launches/step {sum(r[2] for r in rows):.0f}; in-step {tot_in / 1e3:.3f} ms ({tot_fl / tot_in / 1e6:.0f} TFLOP/s), "
      f"alone {tot_al / 1e3:.3f} ms ({tot_fl / tot_al / 1e6:.0f} TFLOP/s), {tot_fl / 1e12:.3f} TFLOP/step")
