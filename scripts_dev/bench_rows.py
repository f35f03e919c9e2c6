"""add_ln forward / backward per shape of the cfg-2 step, chunk-mapped (variant 1) against quad-mapped (variant 0) kernels in ONE
process, interleaved rounds, HIP events around 20 back-to-back launches; bytes = what the kernel moves (fp32 twin in and out)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd  # noqa: F401
from hri_emo_amd import _ops, _lib

d = int(sys.argv[1]) if len(sys.argv) > 1 else 768
shapes = [(25600, "audio"), (8192, "text"), (384, "decoder")]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for M, name in shapes:
    dy = torch.randn(M, d, device="cuda").bfloat16(); g = torch.randn(M, d, device="cuda").bfloat16()
    x32 = torch.randn(M, d, device="cuda")
    gamma = torch.ones(d, device="cuda"); beta = torch.zeros(d, device="cuda")
    res = {0: [[], []], 1: [[], []]}
    for rnd in range(4):
        for v in (1, 0):
            _lib.call("hriemo_rowops_force_variant", v)
            y, y32, mean, rstd = _ops.add_ln_fwd(g, None, gamma, beta, 0.1, 1, 2, 0, x32=x32, want32=True)
            res[v][0].append(timed(lambda: _ops.add_ln_fwd(g, None, gamma, beta, 0.1, 1, 2, 0, x32=x32, want32=True)))
            # partials only (the step's form: the launch-boundary reduce finishes the column sums)
            rows = _lib.lib().hriemo_add_ln_bwd_partial_rows(M, d)
            part = torch.empty(rows * 3 * d, dtype=torch.float32, device="cuda")
            dx = torch.empty_like(g); dg = torch.empty_like(g)
            P = _ops._p
            res[v][1].append(timed(lambda: _lib.call("hriemo_add_ln_bwd_rows", P(dy), P(g), None, P(x32), P(gamma), P(mean), P(rstd), P(dx), P(dg),
                                                     None, None, None, 0, M, d, 0.1, 1, P(_ops.seed_word(g.device)), 2, 0, P(part), None, _ops._stream())))
    _lib.call("hriemo_rowops_force_variant", 1)
    bf, bb = M * d * 12, M * d * 12
    for v, nm in ((1, "chunk"), (0, "quad ")):
        f, b = sorted(res[v][0]), sorted(res[v][1])
        print(f"d={d} {name:8s} M={M:6d} {nm}: fwd median {f[len(f)//2]:7.1f} us min {f[0]:7.1f} ({bf / f[len(f)//2] / 1e6:5.2f} TB/s)   "
              f"bwd median {b[len(b)//2]:7.1f} us min {b[0]:7.1f} ({bb / b[len(b)//2] / 1e6:5.2f} TB/s)", flush=True)
