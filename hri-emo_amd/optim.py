"""clip_grad_norm_(max_norm) + AdamW for the fusion path as two HIP passes over flat buffers
(scripts/fusion/train_fusion_seq_level_decoder.py:332-334).

The gradients already live in ONE flat fp32 buffer (dp.GradBuckets).  This optimizer gives the parameters and both
Adam moments the same layout -- every ``p.data`` becomes a view into a flat parameter buffer -- so the whole update
is one elementwise kernel and the gradient norm one reduction, with the clip coefficient taken from device memory
(no host synchronisation, safe between hipGraph replays).  Construct it BEFORE ``DataParallelStep.capture()``:
it moves the parameter storage.  Results follow torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW (fp32)."""
import torch

from . import _lib
from . import _ops
from ._ops import _p, _stream, _require_gpu


class FusedClipAdamW:
    def __init__(self, buckets, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=5.0):
        self.buckets = buckets
        flat_g = buckets.flat
        _require_gpu(flat_g)
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.flat_p = torch.zeros_like(flat_g)
        self.m = torch.zeros_like(flat_g)
        self.v = torch.zeros_like(flat_g)
        with torch.no_grad():
            for p in buckets.params:
                off, n = buckets._offsets[id(p)], p.numel()
                view = self.flat_p[off:off + n].view_as(p)
                view.copy_(p.data)
                p.data = view                       # same values, storage now inside the flat buffer
        self.nblocks = 1024
        self._partial = torch.empty(self.nblocks, dtype=torch.float32, device=flat_g.device)
        self._norm2 = torch.zeros(1, dtype=torch.float32, device=flat_g.device)
        self.steps = 0

    @torch.no_grad()
    def step(self):
        """One update from the gradients currently in the flat buffer; returns the pre-clip gradient norm (device)."""
        g = self.buckets.flat
        n = g.numel()
        st = _stream()
        self.steps += 1
        _lib.call("hriemo_sumsq_f32", _p(g), n, _p(self._partial), self.nblocks, st)
        _lib.call("hriemo_rowsum_f32", _p(self._partial), _p(self._norm2), 1, self.nblocks, st)
        _lib.call("hriemo_adamw_flat", _p(self.flat_p), _p(g), _p(self.m), _p(self.v), n, self.lr, self.betas[0], self.betas[1],
                  self.eps, self.wd, self.steps, self.max_norm if self.max_norm else 0.0, _p(self._norm2), st)
        # the update went through raw pointers: neither p._version nor p.data_ptr() moved, so tell the bf16 weight
        # shadows explicitly (eager steps would otherwise keep running on the initial weights)
        _ops.bump_weights_epoch()
        return self._norm2.sqrt()
