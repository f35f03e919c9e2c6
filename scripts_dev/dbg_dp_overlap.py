"""Debug aid: two ranks on one GPU over gloo (as tests/test_gpu_parity.py::test_two_ranks_overlapped_allreduce_equals_single_rank).
Each rank computes its local gradients twice -- hooks off (no exchange) and hooks on (overlapped exchange) -- and the exchanged
average is compared per bucket with the mean of the ranks' local gradients: tells a wrong exchange from a wrong local step."""
import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, reps):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hri_emo_amd as H
    from hri_emo_amd.dp import DataParallelStep, GradBuckets
    from hri_emo_amd.train import fusion_step_loss
    torch.cuda.set_device(0)
    g = torch.Generator().manual_seed(31)
    B, Ta, Tt, d = 8, 48, 24, 128
    h_a, h_t = torch.randn(B, Ta, d, generator=g), torch.randn(B, Tt, d, generator=g)
    m_a = torch.arange(Ta)[None] >= torch.randint(Ta // 2, Ta + 1, (B, 1), generator=g)
    m_t = torch.arange(Tt)[None] >= torch.randint(Tt // 2, Tt + 1, (B, 1), generator=g)
    y = (torch.rand(B, 4, generator=torch.Generator().manual_seed(5)) < 0.3).float()
    torch.manual_seed(3)
    m = H.FusionWithEmotionDecoder(d_model=d, num_emotions=4, n_heads=8, dropout=0.0).cuda().train()
    dp = DataParallelStep(m, fusion_step_loss, bucket_bytes=256 << 10, overlap=True)
    lo, hi = dp.set_global_batch(B)
    batch = (h_a[lo:hi].cuda().bfloat16(), h_t[lo:hi].cuda().bfloat16(), m_a[lo:hi].cuda(), m_t[lo:hi].cuda(), y[lo:hi].cuda())
    names = {id(p): n for n, p in m.named_parameters()}
    mode = os.environ.get("DBG_MODE", "")
    if mode == "noop":                                   # exchange replaced by nothing: hooks-on LOCAL gradients vs hooks-off
        class W:
            def wait(self): pass
        dist_all_reduce = dist.all_reduce
        import hri_emo_amd.dp as dpm
        dpm.dist.all_reduce = lambda t, **kw: W()
    for rep in range(reps):
        dp.buckets.suspended = True                      # local gradients, no exchange
        dp._fwd_bwd(*batch)
        torch.cuda.synchronize()
        local = dp.buckets.flat.clone()
        gathered = [torch.empty_like(local.cpu()) for _ in range(world)]
        dist.all_gather(gathered, local.cpu())
        want = sum(gathered) / world
        if mode == "noop":
            want = gathered[rank] / world
        dp.buckets.suspended = False
        dp.step(*batch)
        torch.cuda.synchronize()
        got = dp.buckets.flat.cpu()
        bad = []
        for bi, (s, e, n) in enumerate(dp.buckets.buckets):
            r = float((got[s:e] - want[s:e]).norm() / want[s:e].norm().clamp_min(1e-20))
            rl = float((got[s:e] - gathered[rank][s:e] / world).norm() / want[s:e].norm().clamp_min(1e-20))
            if r > 1e-5:
                g0, g1 = gathered[0][s:e], gathered[1][s:e]
                cands = {"g0/2": g0 / 2, "g1/2": g1 / 2, "g0": g0, "g1": g1, "(g0+g1)": g0 + g1, "g0+g1/2": g0 + g1 / 2, "g0/2+g1": g0 / 2 + g1, "0": g0 * 0}
                best = min(cands, key=lambda k: float((got[s:e] - cands[k]).norm()))
                # per-parameter view: which rows are off
                det = []
                for p in dp.buckets.params:
                    if dp.buckets._bucket_of[id(p)] == bi:
                        o, n = dp.buckets._offsets[id(p)], p.numel()
                        rows = p.shape[0] if p.dim() >= 2 else 1
                        err = (got[o:o + n] - want[o:o + n]).view(rows, -1).norm(dim=1) / (want[o:o + n].view(rows, -1).norm(dim=1) + 1e-20)
                        badrows = (err > 1e-4).nonzero().flatten()
                        det.append((names[id(p)].split("layers.")[-1], int(badrows.numel()), rows, int(badrows.min()) if badrows.numel() else -1, int(badrows.max()) if badrows.numel() else -1))
                bad.append((bi, round(r, 4), best, round(float((got[s:e] - cands[best]).norm() / want[s:e].norm()), 4), det))
        print(f"rank {rank} rep {rep}: {len(bad)} bad buckets", *bad[:8], sep="\n    ", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    ctx = mp.get_context("spawn")
    port = 29900 + os.getpid() % 90
    ps = [ctx.Process(target=worker, args=(r, 2, port, reps)) for r in range(2)]
    for p in ps: p.start()
    for p in ps: p.join()
