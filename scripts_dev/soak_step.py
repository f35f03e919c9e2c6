"""Whole-step soak: the captured cfg-2 step (dropout 0.1) replayed N times from the SAME dropout seed word must give bit-identical
gradients and loss every time -- every kernel sums in a fixed order, so any differing bit is a fault (a race, a lost lane, an
instruction hazard like the one of DESIGN.md 3.2), not noise.  Prints the number of differing replays and elements."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd as H
from hri_emo_amd import _ops
from hri_emo_amd.dp import DataParallelStep
from hri_emo_amd.train import fusion_step_loss
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64          # 16 + ragged masks = the shape of the GPU suite's determinism test
ragged = len(sys.argv) > 3 and sys.argv[3] == "ragged"
dev = torch.device("cuda", 0)
if len(sys.argv) > 4:                                       # perturb the allocator first (the suite runs other tests before)
    junk = [torch.empty(int(x) << 20, dtype=torch.uint8, device=dev) for x in sys.argv[4].split(",")]
    del junk[::2]
torch.manual_seed(1234)
model = H.FusionWithEmotionDecoder(**bench.CFG).to(dev).train()
dp = DataParallelStep(model, fusion_step_loss, overlap=False)
dp.set_global_batch(B)
batch = bench.synth(B, 0, dev)
if ragged:
    g_ = torch.Generator().manual_seed(77)
    la, lt = torch.randint(200, 401, (B,), generator=g_), torch.randint(64, 129, (B,), generator=g_)
    batch = (batch[0], batch[1], (torch.arange(400)[None] >= la[:, None]).to(dev), (torch.arange(128)[None] >= lt[:, None]).to(dev), batch[4])
names = {id(p): nm for nm, p in model.named_parameters()}
if "GEMM_FLAGS" in os.environ:                             # e.g. 1 = the loader / consumer GEMMs draw their tiles from the work queue
    from hri_emo_amd import _lib
    _lib.lib().hriemo_gemm_debug_flags(int(os.environ["GEMM_FLAGS"]))
if os.environ.get("VARLEN", "0") == "1":                   # the packed (varlen) bucket graph: surplus rows must never leak into a sum
    H.set_varlen(True)
dp.step(*batch)
dp.capture(*batch)
sw = _ops.seed_word(dev)
ref = None
bad_replays = bad_elems = 0
for i in range(n):
    sw.fill_(12345)
    loss = dp.step(*batch)
    torch.cuda.synchronize()
    g = dp.buckets.flat.view(torch.int32)
    if ref is None:
        ref, ref_loss = g.clone(), loss.clone()
        print(f"reference replay: loss {float(loss):.6f}, |grad| {float(dp.buckets.flat.norm()):.6f}, {g.numel()} gradient words", flush=True)
        continue
    d = int((g != ref).sum())
    if d or not torch.equal(loss, ref_loss):
        bad_replays += 1
        bad_elems += d
        if bad_replays <= 3:
            idx = (g != ref).nonzero().flatten()[:4].tolist()
            print(f"  replay {i}: {d} differing gradient words (first at {idx}), loss {float(loss):.8f} vs {float(ref_loss):.8f}", flush=True)
            clean = []
            for p in dp.buckets.params:
                o, k = dp.buckets._offsets[id(p)], p.numel()
                if p.dim() >= 2 and not bool((g[o:o + k] != ref[o:o + k]).any()):
                    clean.append(names[id(p)])
            print("    matrices that did NOT change:", clean, flush=True)
pb = getattr(dp, "_pb", None)
packed = f"varlen {_ops.varlen()}" + (f", {len(pb['graphs'])} bucket graph(s) {sorted(pb['graphs'])}" if pb else "")
valid = float((~batch[2]).float().mean() + (~batch[3]).float().mean()) / 2
print(f"STEP SOAK {'CLEAN' if bad_replays == 0 else 'DIRTY'}: {bad_replays} of {n - 1} replays differ ({bad_elems} words) -- "
      f"{(n - 1) * ref.numel():.3g} gradient words compared")
print(f"STEP SOAK configuration: argv {sys.argv[1:]}, B {B}, ragged {ragged} (valid fraction {valid:.3f}), {packed}, "
      f"two streams {_ops.side_stream(dev) is not None}, gemm {_ops.gemm_mode()}, precision {_ops.precision()}, GEMM_FLAGS {os.environ.get('GEMM_FLAGS', 'default (9)')}")
