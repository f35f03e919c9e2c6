"""Host-side batch plumbing of the seq-level trainer (scripts/fusion/train_fusion_seq_level_decoder.py:137-232):
the on-disk feature format {"hidden": [L, d], "attention_mask": [L] (1 = valid)} -> padded batch + key-padding
masks (True = PAD), plus what the reference leaves on the table:

* ``trim_padding`` drops the columns that are PAD for EVERY sample of the batch (feature files are usually stored
  padded to a dataset-wide length, so the reference computes -- and this path would otherwise compute -- all of those
  query rows).  Outputs (logits, beta, z) are unchanged: PAD keys are masked, PAD query rows feed nothing.
* ``length_bucketed_batches`` groups utterances of similar length so a batch's padding (= wasted rows in every
  GEMM / LayerNorm / attention of the path) stays small.
"""
import torch


def load_seq_feat(obj):
    """{"hidden": [L,d], "attention_mask": [L] 1=valid} -> (hidden fp32 [L,d], mask bool [L] True=PAD)   (:137-154)"""
    return obj["hidden"].float(), obj["attention_mask"].long() == 0


def collate_seq_batch(batch, loss_type="multi_label"):
    """list of (h_a[L_a,d], m_a[L_a], h_t[L_t,d], m_t[L_t], label) -> (h_a[B,La,d], mask_a[B,La], h_t[B,Lt,d],
    mask_t[B,Lt], labels); zero padding, padded positions masked True; labels [B] long (single_label) or [B,C]
    float (multi_label)   (:191-232)"""
    B = len(batch)
    d = batch[0][0].shape[-1]
    La = max(s[0].shape[0] for s in batch)
    Lt = max(s[2].shape[0] for s in batch)
    h_a, h_t = torch.zeros(B, La, d), torch.zeros(B, Lt, d)
    m_a, m_t = torch.ones(B, La, dtype=torch.bool), torch.ones(B, Lt, dtype=torch.bool)
    for i, (xa, ka, xt, kt, _) in enumerate(batch):
        h_a[i, :xa.shape[0]], m_a[i, :xa.shape[0]] = xa, ka
        h_t[i, :xt.shape[0]], m_t[i, :xt.shape[0]] = xt, kt
    if loss_type == "single_label":
        labels = torch.tensor([s[4] for s in batch], dtype=torch.long)
    else:
        labels = torch.stack([s[4] for s in batch], dim=0)
    return h_a, m_a, h_t, m_t, labels


def _valid_extent(mask):
    """1 + index of the last position that is valid for at least one sample (0 if everything is PAD)."""
    any_valid = (~mask).any(dim=0)
    idx = torch.nonzero(any_valid)
    return int(idx[-1]) + 1 if idx.numel() else 0


def trim_padding(h_a, mask_a, h_t, mask_t):
    """Cut the trailing columns that are PAD in every sample.  The audio side is never cut below the text side:
    BetaGate fuses over the text length and slices h_a[:, :L_t] (beta_gate_tacfn.py:98-112)."""
    Lt = max(1, _valid_extent(mask_t))
    La = max(1, _valid_extent(mask_a), min(Lt, h_a.shape[1]))
    return h_a[:, :La].contiguous(), mask_a[:, :La].contiguous(), h_t[:, :Lt].contiguous(), mask_t[:, :Lt].contiguous()


def length_bucketed_batches(lengths, batch_size, shuffle=True, generator=None, bucket_mult=50):
    """Index batches in which lengths are similar: shuffle, cut into chunks of bucket_mult*batch_size, sort each chunk
    by length, slice it into batches, shuffle the batches.  Every index appears exactly once."""
    n = len(lengths)
    order = torch.randperm(n, generator=generator).tolist() if shuffle else list(range(n))
    chunk = max(batch_size, bucket_mult * batch_size)
    batches = []
    for s in range(0, n, chunk):
        part = sorted(order[s:s + chunk], key=lambda i: lengths[i])
        batches += [part[k:k + batch_size] for k in range(0, len(part), batch_size)]
    if shuffle:
        perm = torch.randperm(len(batches), generator=generator).tolist()
        batches = [batches[i] for i in perm]
    return batches
