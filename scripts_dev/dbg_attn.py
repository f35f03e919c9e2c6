import os, sys, math, torch
os.environ.setdefault('HRIEMO_ATTN_FUSED_BWD', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hri_emo_amd
from hri_emo_amd import _ops as ops
import hashrng
def run(B, H, Lq, Lk, hd, masked, p, use_bits=True):
    g = torch.Generator().manual_seed(100 + Lq + Lk)
    d = H * hd
    qb = (torch.randn(B * Lq, d, generator=g) * 1.5).bfloat16()
    kvb = torch.randn(B * Lk, 2 * d, generator=g).bfloat16()
    dob = torch.randn(B * Lq, d, generator=g).bfloat16()
    kpm = None
    if masked:
        lens = torch.randint(max(1, Lk // 2), Lk + 1, (B,), generator=g)
        kpm = torch.arange(Lk)[None, :] >= lens[:, None]
    seed, site, boff = 1234567890123, 40, 5
    keep = torch.from_numpy(hashrng.attn_mask(seed, site, B, H, Lq, Lk, p, boff)).float() if p > 0 else None
    q = qb.float().view(B, Lq, H, hd).transpose(1, 2).detach().requires_grad_(True)
    k = kvb[:, :d].float().contiguous().view(B, Lk, H, hd).transpose(1, 2).detach().requires_grad_(True)
    v = kvb[:, d:].float().contiguous().view(B, Lk, H, hd).transpose(1, 2).detach().requires_grad_(True)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if kpm is not None: s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
    pr = torch.softmax(s, -1)
    pd = pr if keep is None else pr * keep * hashrng.inv_keep(p)
    o_ref = (pd @ v).transpose(1, 2).reshape(B * Lq, d)
    o_ref.backward(dob.float())
    qd, kvd, dod = qb.cuda(), kvb.cuda(), dob.cuda()
    kpm_d = kpm.cuda().view(torch.uint8) if kpm is not None else None
    o, lse, mb = ops.attn_fwd(qd, kvd[:, :d], kvd[:, d:], B, H, Lq, Lk, hd, kpm_d, p, seed, site, boff, want_bits=True)
    dq = torch.empty_like(qd); dkv = torch.empty_like(kvd)
    import hri_emo_amd._ops as _o
    _orig_empty_like = torch.empty_like
    from hri_emo_amd import _lib
    L = _lib.lib()
    dbg = None
    if hasattr(L, "hriemo_attn_dbg"):
        import ctypes
        dbg = torch.full((B * H * Lq, 4, 2), float("nan"), device="cuda")
        L.hriemo_attn_dbg.argtypes = [ctypes.c_void_p]; L.hriemo_attn_dbg.restype = None
        L.hriemo_attn_dbg(dbg.data_ptr())
    ops.attn_bwd(qd, kvd[:, :d], kvd[:, d:], o, dod, dq, dkv[:, :d], dkv[:, d:], lse, B, H, Lq, Lk, hd, kpm_d, p, seed, site, boff, mask_bits=mb if use_bits else None)
    delta = ops.attn_bwd.last_delta.cpu()
    if dbg is not None:
        torch.cuda.synchronize()
        dd = dbg.cpu().view(B, H, Lq, 4, 2)
        first, again = dd[..., 0], dd[..., 1]
        bad1 = ((first - delta.view(B, H, Lq, 1)).abs() > 1e-6).nonzero().tolist()
        bad2 = ((again - delta.view(B, H, Lq, 1)).abs() > 1e-6).nonzero().tolist()
        print(f"  sideband delta as READ by the waves: {len(bad1)} wrong first reads, {len(bad2)} wrong re-reads")
        for (b1, h1, q1, w1) in bad1[:24]:
            print(f"    b{b1} h{h1} q{q1} (row {q1 % 32}) wave {w1}: first {float(first[b1,h1,q1,w1]):+.5f} again {float(again[b1,h1,q1,w1]):+.5f} true {float(delta.view(B,H,Lq)[b1,h1,q1]):+.5f}")
    dref = (dob.float() * o.float().cpu()).view(B, Lq, H, hd).sum(-1).permute(0, 2, 1) * (1.0 / hashrng.inv_keep(p) if p > 0 else 1.0)
    de = (delta - dref).abs()
    print(f"  delta: max err {de.max():.4f} (ref max {dref.abs().max():.2f}); rows off by > 0.05: {(de > 0.05).nonzero().tolist()[:12]}")
    refs = dict(dq=q.grad.transpose(1, 2).reshape(B, Lq, H, hd), dk=k.grad.transpose(1, 2).reshape(B, Lk, H, hd), dv=v.grad.transpose(1, 2).reshape(B, Lk, H, hd))
    gots = dict(dq=dq.float().cpu().view(B, Lq, H, hd), dk=dkv[:, :d].float().cpu().view(B, Lk, H, hd), dv=dkv[:, d:].float().cpu().view(B, Lk, H, hd))
    badq = ((gots["dq"] - refs["dq"]).abs().amax(-1) > 3e-2 * max(1.0, refs["dq"].abs().max().item())).nonzero().tolist()   # (b, q, h)
    ik = hashrng.inv_keep(p) if p > 0 else 1.0
    for (b1, q1, h1) in badq[:10]:
        dqe = (gots["dq"][b1, q1, h1] - refs["dq"][b1, q1, h1]).double()                 # [hd] error of the dQ row
        Kh = k[b1, h1].detach().double()                                                    # [Lk, hd]
        coef = (Kh @ dqe) / (Kh * Kh).sum(-1) * math.sqrt(hd)                               # per key: error of dS if it were the only wrong one
        resid = ((dqe[None, :] - coef[:, None] * Kh / math.sqrt(hd)) ** 2).sum(-1)
        kk = int(resid.argmin())
        P = pr[b1, h1, q1].detach().double(); dP = (dob.float().view(B, Lq, H, hd)[b1, q1, h1].double() @ v[b1, h1].detach().double().t())
        kp = keep[b1, h1, q1].double() if keep is not None else torch.ones(Lk, dtype=torch.double)
        delta_row = float((P * kp * ik * dP).sum())
        true_ds = float(P[kk] * (kp[kk] * ik * dP[kk] - delta_row))
        print(f"   b{b1} h{h1} q{q1}: best single key {kk} (resid {float(resid[kk]):.2e} of {float((dqe**2).sum()):.2e}) dS err {float(coef[kk]):+.4f}; keep={int(kp[kk])} "
              f"P={float(P[kk]):.4f} dP={float(dP[kk]):+.3f} delta={delta_row:+.3f} true dS={true_ds:+.4f}; candidates: -P*ik*dP={-float(P[kk]*ik*dP[kk]):+.4f} "
              f"+P*ik*dP={float(P[kk]*ik*dP[kk]):+.4f} P*delta={float(P[kk])*delta_row:+.4f}")
    print(f"case B{B} H{H} Lq{Lq} Lk{Lk} hd{hd} masked={masked} p={p} bits={use_bits} fused_env={os.environ.get('HRIEMO_ATTN_FUSED_BWD')}")
    for n in refs:
        e = (gots[n] - refs[n]).abs()
        bad = (e > 3e-2 * max(1.0, refs[n].abs().max().item()))
        print(f"  {n}: max err {e.max():.3f} (ref max {refs[n].abs().max():.2f}) bad elements {int(bad.sum())} / {bad.numel()}")
        if bad.any():
            idx = bad.nonzero()
            print("    bad (b,row,h):", sorted(set((int(x[0]), int(x[1]), int(x[2])) for x in idx))[:30])
            if kpm is not None: print("    lens of bad b:", [int(lens[b]) for b in sorted(set(idx[:, 0].tolist()))[:20]])
os.environ.setdefault("HRIEMO_ATTN_FUSED_BWD", "1")
for args in [(64, 8, 128, 128, 96, False, 0.1, True), (8, 8, 400, 128, 96, False, 0.1, True)]:
    run(*args)
