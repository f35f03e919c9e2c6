"""Runs every libhriemo_k<N>.so built by make_variants.py on the shapes that showed round 2's dS fault and compares the
single-pass backward (bit-word masks) of each variant with variant k0 (scalar subtract), bit for bit.  Differences are
classified by where they sit: dK rows give the key -> sub-tile kw = (key % 32) / 16 of its wave; dQ rows give the query ->
lane group g = (q % 16) / 4 (lanes 16g .. 16g+15 of the wave that computed dS for it).  Also times each variant.
usage: python scripts_dev/forensics/run_variants.py [reps] [variants...]      (HRIEMO_ATTN_PAIR=1 for the 512-thread pairing)"""
import os, sys, glob, ctypes, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from hri_emo_amd import _lib, _ops as ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ks = [int(x) for x in sys.argv[2:]] or sorted(int(os.path.basename(p)[len("libhriemo_k"):-3]) for p in glob.glob(os.path.join(HERE, "libhriemo_k*.so")))
handles = {}
for k in ks:
    _lib.LIB_PATH = os.path.join(HERE, f"libhriemo_k{k}.so"); _lib._lib = None
    handles[k] = _lib.lib()
def use(k): _lib._lib = handles[k]

SHAPES = [(64, 8, 400, 128, 96), (64, 8, 128, 128, 96)]
print(f"pair={os.environ.get('HRIEMO_ATTN_PAIR', '0')} reps={reps} variants={ks}", flush=True)
for (B, H, Lq, Lk, hd) in SHAPES:
    d = H * hd
    stats = {k: dict(n=0, kw=[0, 0], g=[0, 0, 0, 0], dv=0, runs_bad=0) for k in ks}
    for rep in range(reps):
        gen = torch.Generator().manual_seed(1000 * rep + Lq + Lk + hd)
        qd = (torch.randn(B * Lq, d, generator=gen) * 1.5).bfloat16().cuda()
        kvd = torch.randn(B * Lk, 2 * d, generator=gen).bfloat16().cuda()
        dod = torch.randn(B * Lq, d, generator=gen).bfloat16().cuda()
        seed, site, boff, p = 77 + rep, 40, rep, 0.1
        use(ks[0])
        o, lse, mb = ops.attn_fwd(qd, kvd[:, :d], kvd[:, d:], B, H, Lq, Lk, hd, None, p, seed, site, boff, want_bits=True)
        ref = None
        for k in ks:
            use(k)
            dq = torch.empty_like(qd); dkv = torch.empty_like(kvd)
            ops.attn_bwd(qd, kvd[:, :d], kvd[:, d:], o, dod, dq, dkv[:, :d], dkv[:, d:], lse, B, H, Lq, Lk, hd, None, p, seed, site, boff, mask_bits=mb)
            torch.cuda.synchronize()
            if ref is None:
                ref = (dq, dkv); continue
            s = stats[k]
            bq = (dq.view(torch.int16) != ref[0].view(torch.int16))
            bk = (dkv[:, :d].view(torch.int16) != ref[1][:, :d].view(torch.int16))
            s["dv"] += int((dkv[:, d:].view(torch.int16) != ref[1][:, d:].view(torch.int16)).sum())
            nb = int(bq.sum()) + int(bk.sum())
            s["n"] += nb; s["runs_bad"] += nb > 0
            for r in bq.any(dim=1).nonzero().flatten().tolist():
                s["g"][((r % Lq) % 16) // 4] += 1
            for r in bk.any(dim=1).nonzero().flatten().tolist():
                s["kw"][((r % Lk) % 32) // 16] += 1
    # timing: 30 launches of the backward per variant
    for k in ks:
        use(k)
        dq = torch.empty_like(qd); dkv = torch.empty_like(kvd)
        for _ in range(3):
            ops.attn_bwd(qd, kvd[:, :d], kvd[:, d:], o, dod, dq, dkv[:, :d], dkv[:, d:], lse, B, H, Lq, Lk, hd, None, p, seed, site, boff, mask_bits=mb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.attn_bwd(qd, kvd[:, :d], kvd[:, d:], o, dod, dq, dkv[:, :d], dkv[:, d:], lse, B, H, Lq, Lk, hd, None, p, seed, site, boff, mask_bits=mb)
        e1.record(); torch.cuda.synchronize()
        stats[k]["us"] = e0.elapsed_time(e1) / 30 * 1e3
    print(f"B{B} H{H} Lq{Lq} Lk{Lk} hd{hd}: elements compared per variant {reps * (qd.numel() + kvd.numel()):.3g}")
    for k in ks:
        s = stats[k]
        print(f"  k{k}: {s['us']:7.1f} us  differing dQ/dK elements {s['n']:5d} in {s['runs_bad']:2d}/{reps} runs; dK rows by sub-tile kw0/kw1 {s['kw']}; "
              f"dQ rows by lane group g0..g3 {s['g']}; dV diffs {s['dv']}", flush=True)
