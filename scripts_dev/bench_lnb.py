import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd
from hri_emo_amd import _ops
M, d = 25600, 768
dy = torch.randn(M, d, device="cuda").bfloat16(); g = torch.randn(M, d, device="cuda").bfloat16()
x32 = torch.randn(M, d, device="cuda"); x = x32.bfloat16()
gamma = torch.ones(d, device="cuda")
y, y32, mean, rstd = _ops.add_ln_fwd(g, x, gamma, torch.zeros(d, device="cuda"), 0.1, 1, 2, 0, x32=x32, want32=True)
def run(): return _ops.add_ln_bwd(dy, g, x, gamma, mean, rstd, 0.1, 1, 2, 0, x32=x32)
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
print(os.environ.get("HRIEMO_LNB_CAP"), f"{s.elapsed_time(e)/20*1e3:.1f} us (incl. reduce)")
