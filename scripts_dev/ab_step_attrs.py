"""Same-process A/B of host-side switches (attributes of hri_emo_amd._ops) on the captured cfg-2 step: one capture per variant,
replays interleaved.  usage: ab_step_attrs.py base GATE_TWO_STREAMS=False SMALL_DW_ROWS=8192,GROUP_SMALL_DW=False ...
('base' = the defaults; a variant is a comma-separated list of NAME=python-literal)"""
import ast, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd as H
from hri_emo_amd import _ops
from hri_emo_amd.dp import DataParallelStep
from hri_emo_amd.train import fusion_step_loss
import bench
dev = torch.device("cuda", 0)
variants = sys.argv[1:] or ["base"]


def parse(v):
    return {} if v == "base" else {k: ast.literal_eval(x) for k, x in (kv.split("=", 1) for kv in v.split(","))}


defaults = {k: getattr(_ops, k) for v in variants for k in parse(v)}
steps = {}
for v in variants:
    for k, x in defaults.items(): setattr(_ops, k, x)
    for k, x in parse(v).items(): setattr(_ops, k, x)
    torch.manual_seed(1234)
    model = H.FusionWithEmotionDecoder(**bench.CFG).to(dev).train()
    dp = DataParallelStep(model, fusion_step_loss, overlap=False)
    dp.set_global_batch(64)
    batch = bench.synth(64, 0, dev)
    dp.step(*batch)
    dp.capture(*batch)
    steps[v] = (dp, batch)
for k, x in defaults.items(): setattr(_ops, k, x)
res = {v: [] for v in variants}
for rnd in range(5):
    for v in variants:
        dp, batch = steps[v]
        for _ in range(3): dp.step(*batch)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): dp.step(*batch)
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 30)
for v in variants:
    r = sorted(res[v])
    print(f"{v:48s} median {r[2]:.3f} ms  (" + " ".join(f"{x:.3f}" for x in res[v]) + ")", flush=True)
