"""CPU suite: host-side data plumbing (hri_emo_amd.data) that needs no GPU."""


def test_device_prefetcher_passes_batches_through_in_order_on_cpu():
    """hri_emo_amd.data.DevicePrefetcher on a CPU device (what the gloo rehearsals use): every batch once, in order, host-side
    dtype conversion applied, None entries kept, an empty loader yields nothing"""
    import torch
    from hri_emo_amd.data import DevicePrefetcher
    for depth in (2, 3, 5):
        loader = [(torch.full((2, 3), float(i)), None, torch.tensor([i])) for i in range(5)]
        out = list(DevicePrefetcher(loader, "cpu", depth=depth, dtypes=(torch.bfloat16, None, None)))
        assert len(out) == 5
        for i, b in enumerate(out):
            assert float(b[0][0, 0]) == i and b[0].dtype == torch.bfloat16 and b[1] is None and int(b[2]) == i
    assert list(DevicePrefetcher([], "cpu")) == []
