"""Data-parallel driver for the fusion path: one process per GPU, utterances sharded contiguously across
ranks, ONE exchange step per iteration -- a sum/avg all-reduce of the parameter gradients (RCCL over xGMI
when the backend is "nccl"; gloo on CPU for tests).  The reference has no distributed code;
the contract here is "N-rank step on shards == 1-rank step on the concatenated batch".

MI355X-first layout: every parameter gradient is a VIEW into one flat fp32 buffer, ordered in reverse
parameter order (the order backward produces them: decoder first, fusion layer 0 last).  Buckets are
contiguous slices of that buffer, so a bucket is all-reduced in place -- no flatten/unflatten copies -- and
is launched from a post-accumulate-grad hook as soon as its last gradient has been written, overlapping the
collective with the rest of backward.  xGMI is point-to-point (7 links x ~153 GB/s), so few large buckets
(default 32 MiB) beat many small ones.
"""
import os

import torch
import torch.distributed as dist

# gradient memset on the side stream beside the first kernels of the forward instead of in front of them: measured neutral
# (8.18 vs 8.14-8.19 ms per step on one box), opt-in
_ZERO_BESIDE = os.environ.get("HRIEMO_ZERO_BESIDE", "0") == "1"


class GradBuckets:
    """comm_dtype: torch.float32 (default: the flat buffer is all-reduced in place) or torch.bfloat16: every bucket is cast to a
    bf16 staging buffer (half the bytes on the xGMI links: 109 MB instead of 218 MB per step at cfg 2), summed by the collective in
    bf16 and cast back into the fp32 flat buffer (the 1/N average is applied in fp32).  A bf16 sum over N <= 8 ranks adds
    ~2^-9 relative rounding per addition on top of the bf16 backward that produced the gradients."""

    def __init__(self, params, bucket_bytes=32 << 20, group=None, overlap=True, comm_dtype=torch.float32, force_exchange=False):
        """force_exchange: install the hooks and run the collectives at world size 1 too (a one-rank all-reduce is the identity:
        rehearses the exchange -- also captured inside a hipGraph -- on a one-GPU box)"""
        self.params = [p for p in params if p.requires_grad]
        self.force_exchange = bool(force_exchange) and dist.is_initialized()
        self.group = group
        self.comm_dtype = comm_dtype
        self._stage = None
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # matrices first, in the order backward produces them (decoder first, fusion layer 0 last); every vector
        # parameter (biases, LayerNorm gain/bias) behind them: their gradients are finished by the launch-boundary
        # reduce at the END of backward (_ops._DeferredReduce), so they share the last bucket(s) and the matrix
        # buckets can be all-reduced while backward is still running
        rev = list(reversed(self.params))
        order = [p for p in rev if p.dim() >= 2] + [p for p in rev if p.dim() < 2]
        pad = lambda n: (n + 63) // 64 * 64          # every view starts 256-B aligned (16-B vector stores)
        total = sum(pad(p.numel()) for p in order)
        dev = order[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.buckets = []           # (start, end, n_params)
        self._bucket_of = {}
        self._offsets = {}
        off, b_start, b_n = 0, 0, 0
        for p in order:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            p._hriemo_fused_grad = True          # the kernels may accumulate straight into this view (_ops.GradSink)
            self._offsets[id(p)] = off
            self._bucket_of[id(p)] = len(self.buckets)
            off += pad(n)
            b_n += 1
            if (off - b_start) * 4 >= bucket_bytes:
                self.buckets.append((b_start, off, b_n))
                b_start, b_n = off, 0
        if b_n:
            self.buckets.append((b_start, off, b_n))
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._reported = set()
        self.suspended = False      # True: gradient-ready notifications are ignored (rank-local diagnostic steps: no collectives)
        self._works = []
        self._hooks = []
        if overlap and (self.world > 1 or self.force_exchange):
            from . import _ops
            # an overlapped exchange wants gradients in production order; one predicate per GradBuckets (a second instance --
            # an eval or EMA wrapper -- must not switch the first one's deferral of small weight-gradient GEMMs on or off)
            self._predicate = _ops.register_hook_predicate(lambda: bool(self._hooks) and not self.suspended)
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
                p._hriemo_grad_ready = self._on_grad_sink     # gradients the kernels accumulate in place (_ops.GradSink)

    def close(self):
        """remove the gradient-ready hooks and this instance's predicate (the parameters keep their flat-buffer views)"""
        from . import _ops
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if getattr(self, "_predicate", None) is not None:
            _ops.unregister_hook_predicate(self._predicate)
            self._predicate = None
        for p in self.params:
            if getattr(p, "_hriemo_grad_ready", None) == self._on_grad_sink:
                del p._hriemo_grad_ready

    # -- hooks ---------------------------------------------------------------------------------
    def _launch(self, bi):
        s, e, _ = self.buckets[bi]
        self._launched[bi] = True
        ctx = None
        if self.flat.is_cuda:
            # a bucket holds gradients written on both branch streams: the collective is ordered after the launching stream
            # only, so that stream first waits for the other one.  Eager steps launch from the stream the gradient-ready
            # notification came in on.  (Round 3 also tried "always from the main stream, which first waits for the notifying
            # one"; in the two-rank rehearsal over gloo that form sporadically sent a bucket out with a wrong half of an
            # in-projection gradient -- scripts_dev/dbg_dp_overlap.py.  The experiment predates the fix of the SharedGrad
            # first-touch race (_ops.SharedGrad) and may have been that fault; the form measured correct stays for eager steps.)
            # Inside a capture the launch MUST come from the capture's origin stream: the
            # process group forks its communication stream from the launching one, and a fork off the side stream would be a
            # fork nested in a fork (_ops.fork; hipStreamEndCapture crashes on those).
            from . import _ops
            streams = _ops.branch_streams(self.flat.device)
            cur = torch.cuda.current_stream(self.flat.device)
            if _ops.CAPTURING and _ops.CAPTURE_ORIGIN is not None:
                main = _ops.CAPTURE_ORIGIN
                for st in streams:
                    if st != main:
                        main.wait_stream(st)
                if cur != main:
                    main.wait_stream(cur)
                    ctx = torch.cuda.stream(main)
                    ctx.__enter__()
            else:
                for st in streams:
                    if st != cur:
                        cur.wait_stream(st)
        try:
            if self.comm_dtype == torch.float32:
                self._works.append((dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True), None))
            else:
                if self._stage is None:
                    self._stage = torch.empty(self.flat.numel(), dtype=self.comm_dtype, device=self.flat.device)
                st = self._stage[s:e]
                st.copy_(self.flat[s:e])                    # fp32 -> bf16 on the launching stream, ordered before the collective
                self._works.append((dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group, async_op=True), (s, e)))
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)

    def _on_grad(self, p, from_sink=False):
        """gradient of `p` is final.  Sources: autograd's post-accumulate hook (gradients returned as tensors) and
        _ops.GradSink (gradients the kernels wrote in place).  While a parameter is sink-managed in this backward pass
        only the sink's notification counts -- it comes after the producing kernels were issued, for bias / LayerNorm
        vectors after the launch-boundary reduce -- and every parameter counts once per step."""
        if self.suspended or (getattr(p, "_hriemo_sink_managed", False) and not from_sink) or id(p) in self._reported:
            return
        self._reported.add(id(p))
        bi = self._bucket_of[id(p)]
        self._pending[bi] += 1
        if self._pending[bi] == self.buckets[bi][2] and not self._launched[bi]:
            self._launch(bi)

    def _on_grad_sink(self, p):
        self._on_grad(p, from_sink=True)

    # -- per-step API --------------------------------------------------------------------------
    def _rebind(self):
        for p in self.params:
            off, n = self._offsets[id(p)], p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * 4:
                p.grad = self.flat[off:off + n].view_as(p)

    def zero_grad(self, beside=False):
        """beside=True (CUDA, two streams): the 218 MB memset runs on the side stream beside the first kernels of the forward
        instead of in front of them; wait_zeroed() orders the caller's stream behind it before the first gradient is written"""
        self._zero_event = None
        side = None
        if beside and self.flat.is_cuda:
            from . import _ops
            side = _ops.side_stream(self.flat.device)
        if side is not None and side != torch.cuda.current_stream(self.flat.device):
            from . import _ops
            _ops.fork(side, torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(side):
                self.flat.zero_()
                self._zero_event = torch.cuda.Event()
                self._zero_event.record(side)
        else:
            self.flat.zero_()
        self._rebind()                   # keep the views even if someone set grads to None

    def wait_zeroed(self):
        ev = getattr(self, "_zero_event", None)
        if ev is not None:
            torch.cuda.current_stream(self.flat.device).wait_event(ev)
            self._zero_event = None

    def finish(self):
        """Complete the gradient exchange of this step and average over ranks."""
        if self.world > 1 or self.force_exchange:
            for bi in range(len(self.buckets)):
                if not self._launched[bi]:          # no hooks, or a parameter got no gradient this step
                    self._launch(bi)
            for w, rng in self._works:
                w.wait()
                if rng is not None:                     # summed bf16 bucket back into the fp32 flat buffer
                    self.flat[rng[0]:rng[1]].copy_(self._stage[rng[0]:rng[1]])
            self._works = []
            self._pending = [0] * len(self.buckets)
            self._launched = [False] * len(self.buckets)
            self._reported = set()
            for p in self.params:
                p._hriemo_sink_managed = False
            if self.world > 1:
                self.flat.mul_(1.0 / self.world)

    def grad_norm(self):
        return self.flat.norm()


def shard_bounds(global_batch, rank, world):
    """Contiguous shard [lo, hi) of rank `rank` (rank r gets utterances [r*B/W, (r+1)*B/W))."""
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


class DataParallelStep:
    """fwd -> loss -> bwd -> gradient all-reduce for one shard; mirrors the order of operations of
    train_one_epoch (scripts/fusion/train_fusion_seq_level_decoder.py:310-334) minus the optimizer.

    ``capture()`` records zero-grad + forward + loss + backward of one step into a hipGraph (the eager step
    costs ~10 ms of host time for ~150 launches; a replay costs microseconds).  Dropout stays fresh per
    replay through the device-resident seed word the graph bumps itself (``_ops.seed_word``); bf16 weight
    shadows are re-cast inside the graph so optimizer updates between replays are honoured.  The gradient
    all-reduce runs after the replay on the flat buffer."""

    _capture_streams = {}         # device index -> the one stream every capture of this process records on

    def __init__(self, model, loss_fn, group=None, bucket_bytes=32 << 20, overlap=True, comm_dtype=torch.float32, force_exchange=False):
        self.model, self.loss_fn = model, loss_fn
        self._keep = []
        self._exchange_in_graph = False
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = GradBuckets(model.parameters(), bucket_bytes, group, overlap, comm_dtype, force_exchange)
        self._graph = None
        self._static = None
        self._static_loss = None

    def set_global_batch(self, global_batch):
        lo, hi = shard_bounds(global_batch, self.rank, self.world)
        if hasattr(self.model, "set_batch_offset"):
            self.model.set_batch_offset(lo)
        return lo, hi

    def _fwd_bwd(self, h_a, h_t, m_a, m_t, y, zero=True, scale=None):
        if zero:
            self.buckets.zero_grad(beside=_ZERO_BESIDE)
        logits, beta, _ = self.model(h_a, h_t, m_a, m_t)
        loss = self.loss_fn(logits, beta, y)
        if scale is not None:
            loss = loss * scale
        self.buckets.wait_zeroed()
        loss.backward()
        return loss.detach()

    def step_accumulated(self, micro_batches):
        """Gradient accumulation (train_mosei_fusion_seq_level_decoder.py:387-396: loss / grad_accum, optimizer every grad_accum
        micro-steps) under data parallelism WITHOUT a gradient exchange per micro-step: the flat buffer is zeroed once, every
        micro-batch accumulates into it (the kernels add in place), gradient-ready notifications stay suspended until the LAST
        micro-batch, whose backward launches the bucket all-reduces as usual.  Returns the mean of the micro-losses."""
        k = len(micro_batches)
        was = self.buckets.suspended
        total = None
        for i, mb in enumerate(micro_batches):
            self.buckets.suspended = True if i + 1 < k else was
            l = self._fwd_bwd(*mb, zero=(i == 0), scale=1.0 / k)
            total = l if total is None else total + l
        self.buckets.suspended = was
        self.buckets.finish()
        return total

    def capture(self, h_a, h_t, m_a, m_t, y, collectives=False):
        """Record one step on static copies of the batch tensors; later ``step()`` calls replay it.

        collectives=True: the gradient exchange is captured INSIDE the graph -- every bucket's all-reduce is launched from its
        gradient-ready hook while backward is being recorded (on the capture's origin stream; the process group forks its
        communication stream from there), the waits and the 1/N average close the graph -- so a replay overlaps the exchange
        with backward exactly like the eager mode does, without the host enqueueing ~450 launches per step.  Needs a backend
        whose collectives can be stream-captured (RCCL can); the hooks must be installed (overlap=True) and not suspended."""
        from . import _ops
        if collectives and not self.buckets._hooks:
            raise RuntimeError("capture(collectives=True) needs the gradient-ready hooks: GradBuckets(overlap=True) at world size > 1 "
                               "(or force_exchange=True)")
        if not collectives and self.buckets._hooks and not self.buckets.suspended:
            raise RuntimeError("capture() needs GradBuckets(overlap=False) or buckets.suspended = True, or collectives=True to "
                               "capture the exchange as well")
        if collectives:
            self.buckets.suspended = False
        self._static = [None if t is None else t.clone() for t in (h_a, h_t, m_a, m_t, y)]
        self.release_graph()              # a re-capture replaces the old graph: its pinned buffers go first
        # ONE capture stream per device for every capture of the process: workspaces are keyed by stream, a fresh stream per
        # capture would allocate (and, with a graph alive, retire instead of free) a fresh 64 MB+ set each time
        dev = self._static[0].device
        side = DataParallelStep._capture_streams.get(dev.index)
        if side is None:
            side = DataParallelStep._capture_streams[dev.index] = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up off the default stream, as graph capture wants
            for _ in range(2):
                self._fwd_bwd(*self._static)
                if collectives:
                    self.buckets.finish()         # the warm-up steps exchange eagerly (every rank alike) and leave the buckets reset
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        _ops.CAPTURING = True
        _ops.CAPTURE_ORIGIN = side        # every helper-stream fork of the step must start here (_ops.fork refuses nested forks)
        _ops.begin_step()
        try:
            # capture on the stream the warm-up ran on: its workspaces (keyed by stream) exist already, so nothing the
            # graph points into comes from the graph's private pool or is first sized during capture.
            # thread_local: the RCCL watchdog thread may query events while we capture
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                _ops.bump_seed_word(self._static[0].device)
                self._static_loss = self._fwd_bwd(*self._static)
                if collectives:
                    self.buckets.finish()         # waits on the captured collectives + the average: part of the graph
        finally:
            _ops.CAPTURING = False
            _ops.CAPTURE_ORIGIN = None
        _ops.GRAPHS_ALIVE += 1            # from now on outgrown workspaces are retired, not freed (_ops.workspace)
        # buffers the captured kernels point into (column-sum partials, queued weight-gradient operands) live exactly as long as
        # this graph: they move from the process-wide lists to the graph's owner
        self._keep = _ops._deferred.keep + _ops._small_dw.keep
        _ops._deferred.keep, _ops._small_dw.keep = [], []
        self._packed = _ops.varlen()      # the graph bakes the sequence lengths of THIS batch in (packed rows, cu_seqlens)
        self._mask_seen = {}              # k -> (data_ptr, _version) of the caller's mask tensor last compared with the captured one
        self._graph = graph
        self._replay = True
        self._exchange_in_graph = bool(collectives)
        if collectives:
            self.buckets.suspended = True         # from now on hooks fire only inside replays (they are baked into the graph)

    def release_graph(self):
        """drop the captured step (before a re-capture, or to free its buffers): the graph, the buffers its kernels point into"""
        from . import _ops
        if self._graph is not None:
            torch.cuda.synchronize()
            self._graph = None
            self._keep = []
            _ops.GRAPHS_ALIVE = max(0, _ops.GRAPHS_ALIVE - 1)
            if _ops.GRAPHS_ALIVE == 0:
                del _ops._ws_retired[:]   # no graph points into the outgrown workspaces any more

    def use_graph(self, on):
        """Switch between the captured replay (gradient exchange after it) and eager launches (exchange from the
        gradient-ready hooks during backward, if installed)."""
        self._replay = bool(on) and self._graph is not None
        self.buckets.suspended = self._replay     # eager steps: the hooks drive the exchange; replays: it follows / is inside the graph

    def step(self, h_a, h_t, m_a, m_t, y):
        if self._graph is not None and getattr(self, "_replay", True):
            for k, (s, t) in enumerate(zip(self._static, (h_a, h_t, m_a, m_t, y))):
                if s is not None and t is not None and s.data_ptr() != t.data_ptr():
                    if getattr(self, "_packed", False) and k in (2, 3):
                        # the comparison is a host-device sync: once per distinct caller tensor (address + version), not per step
                        seen = (t.data_ptr(), t._version)
                        if self._mask_seen.get(k) == seen:
                            continue
                        if not torch.equal(s, t.to(s.dtype)):
                            raise RuntimeError("DataParallelStep: this step was captured with packed (varlen) sequences; its graph can "
                                               "only be replayed with the padding masks it was captured with -- use_graph(False) or "
                                               "capture per length pattern")
                        self._mask_seen[k] = seen
                        continue
                    s.copy_(t)
            self._graph.replay()
            if not self._exchange_in_graph:
                self.buckets.finish()
            return self._static_loss
        loss = self._fwd_bwd(h_a, h_t, m_a, m_t, y)
        self.buckets.finish()
        return loss
