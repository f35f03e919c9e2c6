import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch.multiprocessing as mp
import test_gpu_parity as T

def main():
    import hri_emo_amd as H
    from hri_emo_amd.dp import DataParallelStep
    from hri_emo_amd.train import fusion_step_loss
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = 29911
    procs = [ctx.Process(target=T._dp_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    flat2, nb = q.get(timeout=300)
    [p.join(timeout=120) for p in procs]
    torch.manual_seed(3)
    m = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4, n_heads=8, dropout=0.0).cuda().train()
    h_a, h_t, m_a, m_t = T._rand_batch(8, 48, 24, 128, 31)
    y = (torch.rand(8, 4, generator=torch.Generator().manual_seed(5)) < 0.3).float()
    dp = DataParallelStep(m, fusion_step_loss, bucket_bytes=256 << 10, overlap=False)
    dp.set_global_batch(8)
    dp.step(h_a.cuda().bfloat16(), h_t.cuda().bfloat16(), m_a.cuda(), m_t.cuda(), y.cuda())
    ref = dp.buckets.flat.cpu(); got = torch.from_numpy(flat2)
    bad = 0
    for n, p in m.named_parameters():
        off = dp.buckets._offsets[id(p)]; k = p.numel()
        r, g = ref[off:off + k], got[off:off + k]
        e = ((g - r).norm() / r.norm().clamp_min(1e-30)).item()
        if e > 1e-3:
            bad += 1
            print(f"{n:60s} dim {p.dim()} rel {e:.3f} ratio {(g.norm() / r.norm().clamp_min(1e-30)).item():.3f}")
    print("bad", bad)

if __name__ == "__main__":
    main()
