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
import collections
import logging
import os

import torch
import torch.distributed as dist

log = logging.getLogger("hri_emo_amd.dp")

# packed (varlen) graphs: the packed row count of a modality is rounded up to a multiple of (padded rows / this): finer = fewer
# wasted rows, more distinct graphs over a run (DataParallelStep.capture)
_VARLEN_BUCKETS = max(1, int(os.environ.get("HRIEMO_VARLEN_BUCKETS", "64")))
# packed (varlen) graphs kept alive per captured step: one hipGraph + the buffers its kernels point into per distinct
# (audio rows, text rows) bucket pair; beyond the cap the least recently replayed one is released (and re-captured if it returns)
_VARLEN_MAX_GRAPHS = max(1, int(os.environ.get("HRIEMO_VARLEN_MAX_GRAPHS", "48")))


class GradBuckets:
    """comm_dtype: torch.float32 (default: the flat buffer is all-reduced in place) or torch.bfloat16: every bucket is cast to a
    bf16 staging buffer (half the bytes on the xGMI links: 109 MB instead of 218 MB per step at cfg 2), summed by the collective in
    bf16 and cast back into the fp32 flat buffer (the 1/N average is applied in fp32).  A bf16 sum over N <= 8 ranks adds
    ~2^-9 relative rounding per addition on top of the bf16 backward that produced the gradients."""

    def __init__(self, params, bucket_bytes=32 << 20, group=None, overlap=True, comm_dtype=torch.float32, force_exchange=False,
                 launch_from="notify"):
        """force_exchange: install the hooks and run the collectives at world size 1 too (a one-rank all-reduce is the identity:
        rehearses the exchange -- also captured inside a hipGraph -- on a one-GPU box)
        launch_from: the stream an EAGER step launches a bucket's collective from -- "notify": the stream the gradient-ready
        notification came in on (it first waits for the other branch stream); "main": always the model's main stream (which first
        waits for the notifying one and the side stream).  A captured step always uses the capture's origin stream."""
        if launch_from not in ("notify", "main"):
            raise ValueError("GradBuckets: launch_from is 'notify' or 'main'")
        self.launch_from = launch_from
        self._snap = None           # test hook (enable_launch_snapshots): the flat buffer as every collective saw it at launch
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

    # -- test hook: is every collective launched behind the kernels that produce its bucket? ------------
    def enable_launch_snapshots(self, on=True):
        """Test hook.  While enabled, every bucket is copied to a side buffer at the exact point its collective
        is enqueued -- on the launching stream, in front of the all-reduce, eager and inside a capture alike.  At world size 1
        the all-reduce is the identity, so after the step the side buffer must equal the flat gradient buffer bit for bit; a
        collective launched before one of its gradients was produced leaves the words that were written later different
        (``launch_snapshot_mismatches``).  That makes the ORDERING of the exchange testable on one GPU, which the values of a
        one-rank exchange alone are not.  fp32 buckets only.  Enable BEFORE capture() to have the copies recorded in the graph."""
        if on and self.comm_dtype != torch.float32:
            raise ValueError("launch snapshots compare fp32 buckets")
        self._snap = torch.zeros_like(self.flat) if on else None

    def launch_snapshot_mismatches(self):
        """after a finished step at world size 1 -> [(bucket, words that differ, first differing word's parameter name or index)];
        empty = every collective saw the final gradients of its bucket"""
        if self._snap is None:
            raise RuntimeError("enable_launch_snapshots() first")
        if self.world != 1:
            raise RuntimeError("launch snapshots are compared at world size 1 (the all-reduce must be the identity)")
        out = []
        for bi, (s, e, _) in enumerate(self.buckets):
            a, b = self._snap[s:e], self.flat[s:e]
            ne = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
            n = int(ne.sum())
            if n:
                first = s + int(ne.nonzero()[0])
                owner = next((i for i, p in enumerate(self.params)
                              if self._offsets[id(p)] <= first < self._offsets[id(p)] + p.numel()), None)
                out.append((bi, n, owner))
        return out

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
            capturing = _ops.CTX.capturing and _ops.CTX.capture_origin is not None
            if capturing or self.launch_from == "main":
                main = _ops.CTX.capture_origin if capturing else streams[0]
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
            if self._snap is not None:
                # what the collective is about to read, copied on the launching stream directly in front of it: a launch that
                # precedes the production of one of the bucket's gradients shows as a difference to the final buffer
                self._snap[s:e].copy_(self.flat[s:e])
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

    def zero_grad(self):
        self.flat.zero_()
        self._rebind()                   # keep the views even if someone set grads to None

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


def _in_step_context(fn):
    """run a DataParallelStep method with the step's own _ops.StepContext in force"""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        from . import _ops
        with _ops.use_context(self.ctx):
            return fn(self, *a, **k)
    return wrapper


class DataParallelStep:
    """fwd -> loss -> bwd -> gradient all-reduce for one shard; mirrors the order of operations of
    train_one_epoch (scripts/fusion/train_fusion_seq_level_decoder.py:310-334) minus the optimizer.

    ``capture()`` records zero-grad + forward + loss + backward of one step into a hipGraph (the eager step
    costs ~10 ms of host time for ~150 launches; a replay costs microseconds).  Dropout stays fresh per
    replay through the device-resident seed word the graph bumps itself (``_ops.seed_word``); bf16 weight
    shadows are re-cast inside the graph so optimizer updates between replays are honoured.  The gradient
    all-reduce runs after the replay on the flat buffer."""

    _capture_streams = {}         # device index -> the one stream every capture of this process records on

    def __init__(self, model, loss_fn, group=None, bucket_bytes=32 << 20, overlap=True, comm_dtype=torch.float32, force_exchange=False,
                 launch_from="notify"):
        self.model, self.loss_fn = model, loss_fn
        self._keep = []
        self._exchange_in_graph = False
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = GradBuckets(model.parameters(), bucket_bytes, group, overlap, comm_dtype, force_exchange, launch_from)
        from . import _ops
        self.ctx = _ops.StepContext()      # this step's capture flag / packed plan / join scope: not shared with another model's
        self._graph = None
        self._static = None
        self._static_loss = None
        self._pb = None               # packed (varlen) bucket graphs: {(rows audio, rows text): record}, see capture()
        self._pool = None

    def set_global_batch(self, global_batch):
        lo, hi = shard_bounds(global_batch, self.rank, self.world)
        if hasattr(self.model, "set_batch_offset"):
            self.model.set_batch_offset(lo)
        return lo, hi

    def _fwd_bwd(self, h_a, h_t, m_a, m_t, y, zero=True, scale=None):
        from . import _ops
        # collectives launched from the gradient-ready hooks run BESIDE backward (eagerly, or baked into a capture): the loader /
        # consumer GEMMs then draw their tiles from the work queue (a block whose CU a collective holds late draws fewer); with
        # the chip to itself they walk statically, which is 5-8 % faster per launch alone and equal inside the step
        b = self.buckets
        _ops.gemm_contended(bool(b._hooks) and not b.suspended and b.world > 1)
        if zero:
            self.buckets.zero_grad()
        logits, beta, _ = self.model(h_a, h_t, m_a, m_t)
        loss = self.loss_fn(logits, beta, y)
        if scale is not None:
            loss = loss * scale
        if loss.is_cuda and scale is None:
            loss.backward(gradient=_ops.one(loss.device))      # the fused losses skip the multiply by this very tensor
        else:
            loss.backward()
        return loss.detach()

    @_in_step_context
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

    @_in_step_context
    def capture(self, h_a, h_t, m_a, m_t, y, collectives=False, lengths=None):
        """Record one step on static copies of the batch tensors; later ``step()`` calls replay it.

        collectives=True: the gradient exchange is captured INSIDE the graph -- every bucket's all-reduce is launched from its
        gradient-ready hook while backward is being recorded (on the capture's origin stream; the process group forks its
        communication stream from there), the waits and the 1/N average close the graph -- so a replay overlaps the exchange
        with backward exactly like the eager mode does, without the host enqueueing ~450 launches per step.  Needs a backend
        whose collectives can be stream-captured (RCCL can); the hooks must be installed (overlap=True) and not suspended.

        Packed (varlen) mode (``hri_emo_amd.set_varlen(True)`` and both padding masks given): the sequence lengths are DEVICE
        data of the captured step (cu_seqlens buffers refreshed before every replay), only the packed row counts are baked in,
        rounded up to a bucket (1/64 of the padded rows; the surplus rows form one extra all-zero sequence).  ``step()`` then
        serves EVERY batch of these shapes: a batch whose row counts fall into a bucket not seen yet captures that bucket's
        graph on the spot (same static inputs, one memory pool).  ``lengths`` = (audio lengths, text lengths) as host lists /
        CPU tensors spares the device -> host read of the masks' row sums (one sync per step otherwise)."""
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
        self.release_graph()              # a re-capture replaces the old graph(s): their pinned buffers go first
        self._collectives = bool(collectives)
        if _ops.varlen() and m_a is not None and m_t is not None and _ops.precision() == "bf16":
            if collectives and self.world > 1:
                raise RuntimeError("capture(collectives=True) in packed (varlen) mode: ranks meet new buckets at different steps and a "
                                   "bucket's capture runs eager warm-up exchanges -- capture without collectives (exchange after the "
                                   "replay) or run the padded path")
            dev = self._static[0].device
            B, La, Lt = h_a.shape[0], h_a.shape[1], h_t.shape[1]
            self._pb = {"graphs": collections.OrderedDict(), "B": B, "La": La, "Lt": Lt,
                        "cu_a": torch.zeros(B + 2, dtype=torch.int32, device=dev), "cu_t": torch.zeros(B + 2, dtype=torch.int32, device=dev)}
            key = self._packed_key(m_a, m_t, lengths)
            rec = self._packed_graph(key)
            self._graph, self._static_loss, self._keep = rec["graph"], rec["loss"], []
        else:
            self._graph, self._static_loss, self._keep = self._capture_graph(None)
        self._packed = self._pb is not None
        self._mask_seen = {}
        self._replay = True
        self._exchange_in_graph = bool(collectives)
        if collectives:
            self.buckets.suspended = True         # from now on hooks fire only inside replays (they are baked into the graph)

    # ---- packed (varlen) bucket graphs
    def _packed_key(self, m_a, m_t, lengths):
        """host-side: the batch's cu_seqlens written to the static device buffers + the bucket (rows audio, rows text) it needs"""
        pb = self._pb
        B, La, Lt = pb["B"], pb["La"], pb["Lt"]
        if lengths is None:
            # one device -> host read per step; masks must be prefix masks with no empty sequence (the collate's form)
            va, vt = ~m_a.bool(), ~m_t.bool()
            la, lt = va.sum(1), vt.sum(1)
            ok = ((va == (torch.arange(La, device=m_a.device)[None, :] < la[:, None])).all()
                  & (vt == (torch.arange(Lt, device=m_t.device)[None, :] < lt[:, None])).all() & (la > 0).all() & (lt > 0).all())
            host = torch.cat([la, lt, ok.long().view(1)]).tolist()
            if not host[-1]:
                raise RuntimeError("DataParallelStep (packed mode): padding masks must mark a suffix of every row and leave at least "
                                   "one valid position -- use_graph(False) runs such a batch on the padded path")
            la, lt = host[:B], host[B:2 * B]
        else:
            la, lt = [int(x) for x in lengths[0]], [int(x) for x in lengths[1]]
            if len(la) != B or len(lt) != B or min(la) < 1 or min(lt) < 1 or max(la) > La or max(lt) > Lt:
                raise ValueError("DataParallelStep (packed mode): lengths must hold one value in [1, L] per utterance and modality")
        out = []
        for lens, L, buf in ((la, La, pb["cu_a"]), (lt, Lt, pb["cu_t"])):
            g = max(8, (B * L) // _VARLEN_BUCKETS // 8 * 8)
            g = min(g, L)                         # the surplus rows are ONE sequence of at most L rows
            n = sum(lens)
            rows = (n // g + 1) * g               # > n: the extra sequence is never empty
            cu = [0]
            for x in lens:
                cu.append(cu[-1] + x)
            cu.append(rows)
            buf.copy_(torch.tensor(cu, dtype=torch.int32))
            out.append(rows)
        return tuple(out)

    def _packed_graph(self, key):
        from . import _ops
        pb = self._pb
        rec = pb["graphs"].get(key)
        if rec is None:
            while len(pb["graphs"]) >= _VARLEN_MAX_GRAPHS:
                # bounded: a corpus with a wide length spread would otherwise accumulate a graph + its buffers per bucket pair
                old_key, old = pb["graphs"].popitem(last=False)
                torch.cuda.synchronize()
                old.clear()
                _ops.GRAPHS_ALIVE = max(1, _ops.GRAPHS_ALIVE - 1)
                log.info("packed step: released the graph of bucket %s (cap %d)", old_key, _VARLEN_MAX_GRAPHS)
            seqs = (_ops.seq_bucket(pb["cu_a"], pb["B"], pb["La"], key[0]), _ops.seq_bucket(pb["cu_t"], pb["B"], pb["Lt"], key[1]))
            # a bucket met inside step() runs two eager warm-up passes + the capture pass; each draws dropout seeds from torch's
            # CPU generator.  Ranks meet new buckets at different steps, so the generator is put back: its stream stays the one
            # the caller (and every other rank) sees, as _ops.next_seed documents.
            rng = torch.get_rng_state() if self._graph is not None else None
            try:
                graph, loss, keep = self._capture_graph(seqs)
            finally:
                if rng is not None:
                    torch.set_rng_state(rng)
            rec = pb["graphs"][key] = {"graph": graph, "loss": loss, "keep": keep, "seqs": seqs}
            log.info("packed step: captured the graph of bucket (audio rows, text rows) = %s (%d alive)", key, len(pb["graphs"]))
        else:
            pb["graphs"].move_to_end(key)
        return rec

    def _capture_graph(self, seqs):
        """warm-up + capture of one step on self._static; seqs = the packed plans of a bucket graph (or None) -> (graph, loss, keep)"""
        from . import _ops
        collectives = self._collectives
        # ONE capture stream per device for every capture of the process: workspaces are keyed by stream, a fresh stream per
        # capture would allocate (and, with a graph alive, retire instead of free) a fresh 64 MB+ set each time
        dev = self._static[0].device
        side = DataParallelStep._capture_streams.get(dev.index)
        if side is None:
            side = DataParallelStep._capture_streams[dev.index] = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        _ops.CTX.seq_override = seqs
        try:
            with torch.cuda.stream(side):                 # warm-up off the default stream, as graph capture wants
                for _ in range(2):
                    self._fwd_bwd(*self._static)
                    if collectives:
                        self.buckets.finish()         # the warm-up steps exchange eagerly (every rank alike) and leave the buckets reset
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if collectives:
                # The process group's watchdog thread polls the end event of every EAGER collective until it has seen it complete
                # (every 100 ms).  The warm-up exchanges above are complete now, but not necessarily reaped -- and once the capture
                # below pulls the group's communication stream into the capture, ROCm 7.2 answers a query of an event that was
                # recorded on that stream with hipErrorCapturedEvent, which the watchdog turns into std::terminate (sporadic: 1 run
                # in 8 aborted, gpurun_out/cap_8.log of round 4).  Two watchdog periods let it drain its list first.
                import time
                time.sleep(0.25)
            graph = torch.cuda.CUDAGraph()
            if seqs is not None and self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()       # the bucket graphs never run concurrently: one pool for all
            _ops.CTX.capturing = True
            _ops.CTX.capture_origin = side        # every helper-stream fork of the step must start here (_ops.fork refuses nested forks)
            _ops.begin_step()
            try:
                # capture on the stream the warm-up ran on: its workspaces (keyed by stream) exist already, so nothing the
                # graph points into comes from the graph's private pool or is first sized during capture.
                # thread_local: the RCCL watchdog thread may query events while we capture
                kw = {"pool": self._pool} if seqs is not None else {}
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local", **kw):
                    _ops.bump_seed_word(self._static[0].device)
                    loss = self._fwd_bwd(*self._static)
                    if collectives:
                        self.buckets.finish()         # waits on the captured collectives + the average: part of the graph
            finally:
                _ops.CTX.capturing = False
                _ops.CTX.capture_origin = None
        finally:
            _ops.CTX.seq_override = None
        _ops.GRAPHS_ALIVE += 1            # from now on outgrown workspaces are retired, not freed (_ops.workspace)
        # buffers the captured kernels point into (column-sum partials, queued weight-gradient operands) live exactly as long as
        # this graph: they move from the process-wide lists to the graph's owner
        keep = _ops._deferred.keep + _ops._small_dw.keep
        _ops._deferred.keep, _ops._small_dw.keep = [], []
        return graph, loss, keep

    def release_graph(self):
        """drop the captured step (before a re-capture, or to free its buffers): the graph, the buffers its kernels point into"""
        from . import _ops
        if self._graph is not None:
            torch.cuda.synchronize()
            n = len(self._pb["graphs"]) if self._pb is not None else 1
            self._graph = None
            self._keep = []
            self._pb = None
            _ops.GRAPHS_ALIVE = max(0, _ops.GRAPHS_ALIVE - n)
            if _ops.GRAPHS_ALIVE == 0:
                del _ops._ws_retired[:]   # no graph points into the outgrown workspaces any more

    def use_graph(self, on):
        """Switch between the captured replay (gradient exchange after it) and eager launches (exchange from the
        gradient-ready hooks during backward, if installed)."""
        self._replay = bool(on) and self._graph is not None
        self.buckets.suspended = self._replay     # eager steps: the hooks drive the exchange; replays: it follows / is inside the graph

    @_in_step_context
    def step(self, h_a, h_t, m_a, m_t, y, lengths=None):
        if self._graph is not None and getattr(self, "_replay", True):
            graph, loss = self._graph, self._static_loss
            if self._pb is not None:
                # packed mode: any padding masks of the captured shapes; the lengths travel as device data (cu_seqlens), the row
                # counts pick the bucket graph (captured now if this bucket was not seen before)
                if m_a is None or m_t is None:
                    raise RuntimeError("DataParallelStep (packed mode): both padding masks are needed")
                seen = (m_a.data_ptr(), m_a._version, m_t.data_ptr(), m_t._version)
                if lengths is None and self._mask_seen.get("key") == seen:
                    key = self._mask_seen["val"]          # the same mask tensors as last step: no second device -> host read
                    self._pb["cu_a"].copy_(self._mask_seen["cu"][0]); self._pb["cu_t"].copy_(self._mask_seen["cu"][1])
                else:
                    key = self._packed_key(m_a, m_t, lengths)
                    # the entry HOLDS the masks: while it is the cache entry their addresses cannot be handed to the next batch's
                    # masks (same address + version 0 would otherwise pass for "the same tensors" with other lengths inside)
                    self._mask_seen = {"key": seen, "val": key, "cu": (self._pb["cu_a"].clone(), self._pb["cu_t"].clone()),
                                       "masks": (m_a, m_t)}
                for s, t in zip(self._static, (h_a, h_t, m_a, m_t, y)):
                    if s is not None and t is not None and s.data_ptr() != t.data_ptr():
                        s.copy_(t)
                rec = self._packed_graph(key)             # captures on self._static (already this batch) if the bucket is new
                graph, loss = rec["graph"], rec["loss"]
            else:
                for s, t in zip(self._static, (h_a, h_t, m_a, m_t, y)):
                    if s is not None and t is not None and s.data_ptr() != t.data_ptr():
                        s.copy_(t)
            graph.replay()
            if not self._exchange_in_graph:
                self.buckets.finish()
            return loss
        loss = self._fwd_bwd(h_a, h_t, m_a, m_t, y)
        self.buckets.finish()
        return loss
