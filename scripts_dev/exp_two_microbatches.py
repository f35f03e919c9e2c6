"""EXPERIMENT (timing only, results not checked): would two half-batches pipelined against each other hide the decoder's serial
tail?  Two independent models / captured steps at B=32, replayed concurrently on two streams, against one step at B=64."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd as H
from hri_emo_amd.dp import DataParallelStep
from hri_emo_amd.train import fusion_step_loss
import bench
dev = torch.device("cuda", 0)
torch.manual_seed(1234)

def make(B):
    m = H.FusionWithEmotionDecoder(**bench.CFG).to(dev).train()
    dp = DataParallelStep(m, fusion_step_loss, overlap=False)
    dp.set_global_batch(B)
    batch = bench.synth(B, 0, dev)
    dp.step(*batch); dp.capture(*batch)
    return dp, batch

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

dp64, b64 = make(64)
t64 = timeit(lambda: dp64.step(*b64))
print(f"one step B=64: {t64:.3f} ms -> {64 / t64 * 1e3:.0f} utt/s", flush=True)
dpa, ba = make(32)
dpb, bb = make(32)
t32 = timeit(lambda: dpa.step(*ba))
print(f"one step B=32 alone: {t32:.3f} ms -> {32 / t32 * 1e3:.0f} utt/s", flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1): dpa._graph.replay()
    with torch.cuda.stream(s2): dpb._graph.replay()
for stagger in (False, True):
    if stagger:       # offset the second stream by half a step so that tails meet bodies
        with torch.cuda.stream(s2): dpb._graph.replay()
    tb = timeit(both)
    print(f"two B=32 steps concurrently (stagger={stagger}): {tb:.3f} ms per pair -> {64 / tb * 1e3:.0f} utt/s", flush=True)
