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


def test_device_prefetcher_hands_sequence_lengths_over_from_the_host_masks():
    """mask_slots: the valid lengths of every utterance, counted on the host copy of the padding masks, travel with the batch
    (DataParallelStep.step(..., lengths=) in packed mode: no device -> host read of the masks)"""
    import torch
    from hri_emo_amd.data import DevicePrefetcher
    la, lt = torch.tensor([5, 2, 7]), torch.tensor([3, 3, 1])
    m_a, m_t = torch.arange(7)[None] >= la[:, None], torch.arange(3)[None] >= lt[:, None]
    loader = [(torch.zeros(3, 7, 4), torch.zeros(3, 3, 4), m_a, m_t, torch.zeros(3, 2))] * 2
    for b in DevicePrefetcher(loader, "cpu", mask_slots=(2, 3)):
        assert len(b) == 6 and b[5] == ([5, 2, 7], [3, 3, 1]) and torch.equal(b[2], m_a)
