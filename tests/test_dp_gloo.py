"""CPU suite: the data-parallel driver (hri-emo_amd/dp.py) with world_size 2 over gloo.
Contract: an N-rank step on contiguous shards == a 1-rank step on the concatenated batch.
The compute under the driver is the CPU oracle here (tests may use it); on MI355X it is the HIP path."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_q, mode="plain"):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import hri_emo_amd  # noqa: F401
    from hri_emo_amd.dp import DataParallelStep, shard_bounds
    from hri_emo_amd.train import fusion_step_loss
    from oracle import hri_emo_oracle as O

    torch.manual_seed(7)
    model = O.FusionWithEmotionDecoder(d_model=64, num_emotions=3, n_heads=4, dropout=0.0).train()
    B, Ta, Tt = 8, 12, 6
    g = torch.Generator().manual_seed(3)
    h_a, h_t = torch.randn(B, Ta, 64, generator=g), torch.randn(B, Tt, 64, generator=g)
    m_a = torch.arange(Ta)[None] >= torch.randint(Ta // 2, Ta + 1, (B, 1), generator=g)
    m_t = torch.arange(Tt)[None] >= torch.randint(Tt // 2, Tt + 1, (B, 1), generator=g)
    y = (torch.rand(B, 3, generator=g) < 0.3).float()
    comm = torch.bfloat16 if mode == "bf16" else torch.float32
    dp = DataParallelStep(model, fusion_step_loss, bucket_bytes=64 << 10, overlap=True, comm_dtype=comm)   # several buckets
    lo, hi = dp.set_global_batch(B)
    assert (lo, hi) == shard_bounds(B, rank, world)
    for _ in range(2):                                   # two steps: buckets reset correctly
        if mode == "accum":                              # two micro-batches per rank, ONE exchange (gradient accumulation)
            mid = (lo + hi) // 2
            # mean-reduced losses: each micro-batch holds half of this rank's utterances, so its loss is scaled by 1/2
            loss = dp.step_accumulated([(h_a[a:b], h_t[a:b], m_a[a:b], m_t[a:b], y[a:b]) for a, b in ((lo, mid), (mid, hi))])
        else:
            loss = dp.step(h_a[lo:hi], h_t[lo:hi], m_a[lo:hi], m_t[lo:hi], y[lo:hi])
    grads = {n: p.grad.clone().numpy() for n, p in model.named_parameters()}   # by value through the queue
    flat_ok = all(p.grad.data_ptr() >= dp.buckets.flat.data_ptr() for p in model.parameters())
    if rank == 0:
        out_q.put((grads, float(loss), len(dp.buckets.buckets), flat_ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["plain", "bf16", "accum"])
def test_two_rank_step_equals_single_rank_step_on_concatenated_batch(mode):
    """plain: fp32 buckets; bf16: gradients cross the wire as bf16 (summed in bf16, averaged in fp32); accum: two micro-batches
    per rank accumulate locally and are exchanged once (train_mosei_fusion_seq_level_decoder.py:387-396)."""
    sys.path.insert(0, REPO)
    from oracle import hri_emo_oracle as O
    from hri_emo_amd.train import fusion_step_loss
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + {"plain": 0, "bf16": 7, "accum": 13}[mode]) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    grads, loss0, nbuckets, flat_ok = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert nbuckets > 1 and flat_ok

    torch.manual_seed(7)
    model = O.FusionWithEmotionDecoder(d_model=64, num_emotions=3, n_heads=4, dropout=0.0).train()
    B, Ta, Tt = 8, 12, 6
    g = torch.Generator().manual_seed(3)
    h_a, h_t = torch.randn(B, Ta, 64, generator=g), torch.randn(B, Tt, 64, generator=g)
    m_a = torch.arange(Ta)[None] >= torch.randint(Ta // 2, Ta + 1, (B, 1), generator=g)
    m_t = torch.arange(Tt)[None] >= torch.randint(Tt // 2, Tt + 1, (B, 1), generator=g)
    y = (torch.rand(B, 3, generator=g) < 0.3).float()
    logits, beta, _ = model(h_a, h_t, m_a, m_t)
    fusion_step_loss(logits, beta, y).backward()
    for n, p in model.named_parameters():
        ref = p.grad
        err = (torch.from_numpy(grads[n]) - ref).abs().max().item()
        tol = 1e-2 if mode == "bf16" else 1e-5          # bf16 on the wire: 2^-8 per value, two ranks
        assert err <= tol * max(1.0, ref.abs().max().item()) + 1e-7, (mode, n, err)


def test_shard_bounds():
    from hri_emo_amd.dp import shard_bounds
    assert [shard_bounds(512, r, 8) for r in (0, 3, 7)] == [(0, 64), (192, 256), (448, 512)]
    with pytest.raises(ValueError):
        shard_bounds(10, 0, 4)
