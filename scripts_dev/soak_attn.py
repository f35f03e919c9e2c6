"""Soak of the single-pass attention backward (HRIEMO_ATTN_FUSED_BWD=1): keep-mask-bits variant against the hash variant of the
same kernel, bit for bit, over many seeds at the shapes of cfg 2 (two waves per SIMD / paired workgroups), plus one fp32
reference check per shape.  The sporadic dS fault of round 2 showed up as ~1e-5 of the elements per run; a clean soak is
>= 1e9 compared elements without a single differing one."""
import os, sys, math, torch
os.environ["HRIEMO_ATTN_FUSED_BWD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hri_emo_amd import _ops as ops
SHAPES = [(64, 8, 400, 128, 96), (64, 8, 128, 128, 96), (32, 8, 400, 128, 128), (64, 8, 128, 64, 96), (16, 16, 333, 100, 64), (64, 8, 50, 128, 96)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
total = 0; diff = 0
for (B, H, Lq, Lk, hd) in SHAPES:
    d = H * hd
    nd = 0
    for rep in range(reps):
        g = torch.Generator().manual_seed(1000 * rep + Lq + Lk + hd)
        qd = (torch.randn(B * Lq, d, generator=g) * 1.5).bfloat16().cuda()
        kvd = torch.randn(B * Lk, 2 * d, generator=g).bfloat16().cuda()
        dod = torch.randn(B * Lq, d, generator=g).bfloat16().cuda()
        kpm_d = None
        if rep % 3 == 2:
            lens = torch.randint(max(1, Lk // 2), Lk + 1, (B,), generator=g)
            kpm_d = (torch.arange(Lk)[None, :] >= lens[:, None]).cuda().view(torch.uint8)
        seed, site, boff, p = 77 + rep, 40, rep, 0.1
        o, lse, mb = ops.attn_fwd(qd, kvd[:, :d], kvd[:, d:], B, H, Lq, Lk, hd, kpm_d, p, seed, site, boff, want_bits=True)
        outs = []
        for bits in (mb, None):
            dq = torch.empty_like(qd); dkv = torch.empty_like(kvd)
            ops.attn_bwd(qd, kvd[:, :d], kvd[:, d:], o, dod, dq, dkv[:, :d], dkv[:, d:], lse, B, H, Lq, Lk, hd, kpm_d, p, seed, site, boff, mask_bits=bits)
            outs.append((dq, dkv))
        n = int((outs[0][0].view(torch.int16) != outs[1][0].view(torch.int16)).sum()) + int((outs[0][1].view(torch.int16) != outs[1][1].view(torch.int16)).sum())
        nd += n; total += qd.numel() + kvd.numel()
    diff += nd
    print(f"B{B} H{H} Lq{Lq} Lk{Lk} hd{hd}: {reps} runs, {nd} differing elements (bits vs hash)", flush=True)
print(f"SOAK {'CLEAN' if diff == 0 else 'DIRTY'}: {diff} differing of {total:.3g} compared elements")
