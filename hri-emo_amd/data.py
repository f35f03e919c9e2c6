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


def collate_seq_batch(batch, loss_type="multi_label", pad_to=None):
    """list of (h_a[L_a,d], m_a[L_a], h_t[L_t,d], m_t[L_t], label) -> (h_a[B,La,d], mask_a[B,La], h_t[B,Lt,d],
    mask_t[B,Lt], labels); zero padding, padded positions masked True; labels [B] long (single_label) or [B,C]
    float (multi_label)   (:191-232).  pad_to=(La, Lt): pad to these lengths instead of the batch maxima -- every batch then
    has one shape, which is what a captured step wants; in packed (varlen) mode the extra PAD rows cost nothing
    (DataParallelStep.capture)."""
    B = len(batch)
    d = batch[0][0].shape[-1]
    La = max(s[0].shape[0] for s in batch)
    Lt = max(s[2].shape[0] for s in batch)
    if pad_to is not None:
        if pad_to[0] < La or pad_to[1] < Lt:
            raise ValueError(f"collate_seq_batch: pad_to={tuple(pad_to)} is shorter than this batch's longest sequences ({La}, {Lt})")
        La, Lt = int(pad_to[0]), int(pad_to[1])
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


class DevicePrefetcher:
    """Host -> device hand-over of the batches of a loader, overlapped with the step that runs on the previous batch.

    The reference copies every batch synchronously at the top of its step (``.to(device)``,
    scripts/fusion/train_fusion_seq_level_decoder.py:306-308): at cfg 2 that is 52 MB per batch, 1-4 ms of PCIe time in front
    of an 8 ms step.  Here a batch is staged in pinned host buffers (a ring of ``depth`` sets, allocated once per shape) and
    copied on a dedicated copy stream while the caller's stream still computes on the previous batch; ``__next__`` makes the
    caller's stream wait for the copy's event and hands out device tensors.  A set of staging and device buffers is reused
    only after the step that consumed it has been enqueued ``depth`` batches ago AND its copy event / the consumer stream's
    event recorded at hand-out have completed, so the ring is safe for any consumer that stays on the stream it called
    ``__next__`` on.

    ``convert`` maps the loader's item to a tuple of tensors (None entries pass through); ``dtypes`` optionally casts on the
    host before the copy (bf16 features halve the PCIe bytes).  On a CPU ``device`` the batches pass through unchanged
    (tests, gloo rehearsals).  ``mask_slots=(i, j)``: the tuple entries that are the audio / text padding masks -- the batch is
    handed out with one more element, ``(audio lengths, text lengths)`` as Python lists counted on the HOST copy of the masks, which
    is what ``DataParallelStep.step(..., lengths=)`` wants in packed (varlen) mode instead of reading the masks back from the
    device."""

    def __init__(self, loader, device, depth=2, convert=None, dtypes=None, mask_slots=None):
        self.loader, self.device, self.depth = loader, torch.device(device), max(2, int(depth))
        self.convert, self.dtypes, self.mask_slots = convert, dtypes, mask_slots
        self.cuda = self.device.type == "cuda"
        self._ring = [None] * self.depth          # slot -> {"host": [...], "dev": [...], "copied": event, "released": event}
        self._copy_stream = torch.cuda.Stream(device=self.device) if self.cuda else None

    def __len__(self):
        return len(self.loader)

    def _stage(self, slot, item):
        tensors = self.convert(item) if self.convert is not None else tuple(item)
        if self.dtypes is not None:
            tensors = tuple(t if (t is None or dt is None) else t.to(dt) for t, dt in zip(tensors, self.dtypes))
        extra = ()
        if self.mask_slots is not None:
            extra = (tuple((~tensors[i].bool()).sum(1).tolist() for i in self.mask_slots),)
        if not self.cuda:
            return tensors + extra, None
        ent = self._ring[slot]
        shapes = [None if t is None else (tuple(t.shape), t.dtype) for t in tensors]
        if ent is None or ent["shapes"] != shapes:
            ent = {"shapes": shapes,
                   "host": [None if t is None else torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in tensors],
                   "dev": [None if t is None else torch.empty(t.shape, dtype=t.dtype, device=self.device) for t in tensors],
                   "copied": torch.cuda.Event(), "released": None}
            self._ring[slot] = ent
        if ent["released"] is not None:
            ent["released"].synchronize()          # the step that read this slot's device buffers has finished
        src = []
        for h, t in zip(ent["host"], tensors):
            if t is None or t.is_pinned():         # a loader with pin_memory=True hands pinned tensors over: copied from there
                src.append(t)
            else:
                h.copy_(t)                         # pageable -> pinned staging (host memcpy)
                src.append(h)
        ent["src"] = src                           # referenced until the slot is staged again: the copy below is asynchronous
        with torch.cuda.stream(self._copy_stream):
            for h, d_ in zip(src, ent["dev"]):
                if h is not None:
                    d_.copy_(h, non_blocking=True)
            ent["copied"].record(self._copy_stream)
        return tuple(ent["dev"]) + extra, ent

    def __iter__(self):
        it = iter(self.loader)
        pending = []                               # staged batches, oldest first
        k = 0

        def stage_next():
            nonlocal k
            try:
                item = next(it)
            except StopIteration:
                return
            pending.append(self._stage(k % self.depth, item))
            k += 1

        for _ in range(self.depth - 1):
            stage_next()
        while pending:
            batch, ent = pending.pop(0)
            if ent is not None:
                torch.cuda.current_stream(self.device).wait_event(ent["copied"])
            yield batch
            # resumed when the caller asks for the next batch, i.e. AFTER it has enqueued its step on this one: the host-side
            # staging of a later batch (pageable -> pinned memcpy) and its copy now run beside that step on the GPU
            if ent is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                ent["released"] = ev
            stage_next()
