"""Debug: gradients of one eager DataParallelStep.step with the small-gradient queue on / off, repeated (same seeds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd as H
from hri_emo_amd import _ops
from hri_emo_amd.dp import DataParallelStep
from hri_emo_amd.train import fusion_step_loss
dev = torch.device("cuda", 0)
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
kw = dict(d_model=256, num_emotions=5, n_heads=8, dropout=p)
g = torch.Generator().manual_seed(77)
B, Ta, Tt, d = 4, 90, 40, 256
ha, ht = torch.randn(B, Ta, d, generator=g).to(dev), torch.randn(B, Tt, d, generator=g).to(dev)
ma = torch.zeros(B, Ta, dtype=torch.bool); mt = torch.zeros(B, Tt, dtype=torch.bool)
ma[1, 70:] = True; mt[2, 30:] = True
ma, mt = ma.to(dev), mt.to(dev)
y = (torch.rand(B, 5, generator=torch.Generator().manual_seed(3)) < 0.3).float().to(dev)
runs = []
for on in (True, True, False, False):
    _ops.DEFER_SMALL_DW = on
    torch.manual_seed(11)
    _ops.seed_word(dev).zero_()
    m = H.FusionWithEmotionDecoder(**kw).to(dev).train()
    dp = DataParallelStep(m, fusion_step_loss, overlap=False)
    log = []
    _ops.DROP_LOG = log
    dp.step(ha, ht, ma, mt, y)
    torch.cuda.synchronize()
    _ops.DROP_LOG = None
    runs.append((on, {n: q.grad.detach().clone() for n, q in m.named_parameters()}, log, int(_ops.seed_word(dev).item())))
    print("run", on, "sites", len(log), "first seeds", [e[1] for e in log[:3]], "seed word after", runs[-1][3], flush=True)
for i, j in ((0, 1), (2, 3), (0, 2)):
    worst = sorted(((runs[i][1][n] - runs[j][1][n]).abs().max().item() / max(1e-12, runs[j][1][n].abs().max().item()), n) for n in runs[i][1])[-4:]
    print(f"runs {i} ({runs[i][0]}) vs {j} ({runs[j][0]}): worst relative-to-max differences", [(round(a, 6), n) for a, n in worst], flush=True)
