"""Host side of the kernels: thin tensor->pointer wrappers over the C ABI (include/hriemo.h) and the
``torch.autograd.Function``s the mirrored nn.Modules (``models/``) are built from.

One Function per residual sub-layer of the reference (each is "LayerNorm(x + dropout(sub(x)))"):
  SelfAttnLN   models/cross_modal_block_tacfn.py:74-81,85-92 ; models/emotion_decoder.py:42-43
  CrossAttnLN  models/cross_modal_block_tacfn.py:98-105,111-118 ; models/emotion_decoder.py:48-55
  FFNLN        models/cross_modal_block_tacfn.py:106,119 ; models/emotion_decoder.py:58-59
  BetaGateFn   models/beta_gate_tacfn.py:68-118
  ExpandFn / RowDotFn  models/emotion_decoder.py:127,155
torch is used for device memory, streams and autograd bookkeeping only; all arithmetic on the path is in
libhriemo.so.  Activations are bf16, statistics/parameter gradients fp32.  No CPU fallback exists.
"""
import itertools
import math
import os as _os

import threading

import torch
import torch.nn.functional as F

from . import _lib

BF16 = torch.bfloat16
_EPS = 1e-5
_apply_tls = threading.local()


# ----------------------------------------------------------------------------- plumbing
def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _require_fp32_master(p):
    """Parameters stay fp32 (what AMP / the reference trainer keeps): the kernels read biases and LayerNorm affines as
    fp32 and cast weights to their bf16 shadows themselves.  model.half() / model.bfloat16() is refused loudly."""
    if p.dtype != torch.float32:
        raise TypeError(f"hri_emo_amd: parameter of dtype {p.dtype}; keep the module's parameters in fp32 "
                        "(the bf16 compute copies are made internally, as under torch.autocast)")


def _require_fp32_masters(*ps):
    for p in ps:
        if p is not None:
            _require_fp32_master(p)


def _require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError(
            "hri_emo_amd runs on MI355X only: got a CPU tensor. There is no CPU fallback for this path "
            "(move the module and its inputs to 'cuda').")


_ws_cache = {}
_ws_retired = []           # outgrown workspaces a captured hipGraph may still point into: never freed
GRAPHS_ALIVE = 0           # number of captured steps that bake workspace pointers (dp.DataParallelStep.capture)


def workspace(nbytes, device, slot=0):
    """Caller-owned scratch for the C ABI (split-K slabs, column-sum partials); grows monotonically.
    One per (device, slot, stream): the audio and text branches may run on two streams concurrently.
    Once a step has been captured into a hipGraph the pointers of the workspaces it used are frozen inside the graph:
    an outgrown buffer is then retired (kept alive), never handed back to the allocator."""
    key = (device.index, slot, torch.cuda.current_stream(device).cuda_stream)
    t = _ws_cache.get(key)
    if t is None or t.numel() * 4 < nbytes:
        if t is not None and GRAPHS_ALIVE > 0:
            _ws_retired.append(t)
        n = max(int(nbytes), 64 << 20) // 4 + 16
        t = torch.empty(n, dtype=torch.float32, device=device)
        _ws_cache[key] = t
    return t


_side_streams = {}
TWO_STREAMS = _os_env_flag = None


def side_stream(device):
    """Second stream for the text branch of a CrossModalBlock (independent of the audio branch between the
    joins); None when disabled (HRIEMO_TWO_STREAMS=0)."""
    global TWO_STREAMS
    if TWO_STREAMS is None:
        import os
        TWO_STREAMS = os.environ.get("HRIEMO_TWO_STREAMS", "1") != "0"
    if not TWO_STREAMS:
        return None
    s = _side_streams.get(device.index)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _side_streams[device.index] = s
    return s


class StepContext:
    """State of ONE model's step in flight -- what used to be module globals, so that two models (or two DataParallelSteps) in one
    process cannot see each other's capture flag, packed plan or gradient-join scope.  dp.DataParallelStep owns one and installs
    it around capture() / step() (use_context); models used on their own run on the module's default context.
      capturing      set by DataParallelStep.capture(): weight shadows are re-cast inside the graph, buffers are kept, not freed
      capture_origin the stream the capture runs on (every helper-stream fork must start there: fork())
      seq_override   (Seq audio, Seq text) injected for the bucketed packed graphs (models/cross_modal_block_tacfn.py)
      join_scope     > 0 inside a forward whose outputs ALL depend on both encoder branches (grad_join)
      half_reports   id(parameter) -> SharedProjFn nodes that have written their half of its gradient in this step"""
    __slots__ = ("capturing", "capture_origin", "seq_override", "join_scope", "half_reports")

    def __init__(self):
        self.capturing, self.capture_origin, self.seq_override, self.join_scope, self.half_reports = False, None, None, 0, {}


CTX = StepContext()        # the context in force


class use_context:
    """`with use_context(ctx):` -- ctx is the step context in force inside the block (re-entrant: the previous one comes back)"""

    def __init__(self, ctx):
        self.ctx, self.prev = ctx, None

    def __enter__(self):
        global CTX
        self.prev, CTX = CTX, self.ctx
        return self.ctx

    def __exit__(self, *exc):
        global CTX
        CTX = self.prev
        return False




def fork(child, parent):
    """`child` starts to depend on `parent`: a stream fork.  While a step is being captured, every fork has to start at the
    capture's origin stream: a helper stream forked from an already forked stream (a fork nested inside a fork) makes
    hipStreamEndCapture segfault on ROCm 7.2 (gpurun_out/seg.log of round 2) -- raw HIP events or torch streams
    alike -- so that shape is refused here, as a Python exception, before anything reaches the runtime."""
    if CTX.capturing and CTX.capture_origin is not None and parent != CTX.capture_origin and child != CTX.capture_origin:
        raise RuntimeError("hri_emo_amd: a stream was forked from a stream that is itself a fork of the capture stream; ROCm 7.2 "
                           "crashes in hipStreamEndCapture on nested forks -- fork helper streams from the capturing stream only "
                           "(join the side stream back first), or run this step eagerly")
    child.wait_stream(parent)


_main_streams = {}


def note_main_stream(stream):
    """the stream a fusion block was entered on (its audio branch runs there, the text branch on the side stream)"""
    _main_streams[stream.device.index] = stream


def branch_streams(device):
    """streams the fusion blocks run on: the main one (as last noted) and the text-branch side stream if it exists"""
    out = [_main_streams.get(device.index, torch.cuda.current_stream(device))]
    side = _side_streams.get(device.index)
    if side is not None:
        out.append(side)
    return out


def share(t, stream):
    """tensor produced on another stream is about to be read on `stream`"""
    if isinstance(t, torch.Tensor):
        t.record_stream(stream)
    elif isinstance(t, tuple):
        for e in t:
            share(e, stream)
    elif t is not None and hasattr(t, "cu"):             # Seq (packed sequences): its index tensors
        t.cu.record_stream(stream)
        t.idx.record_stream(stream)
    return t


_site_counter = itertools.count(1)


def new_site_base():
    """Unique dropout-site id base for a module instance (3 sites per sub-layer)."""
    return next(_site_counter) * 4


_seed_words = {}
STEP_ID = 0                # bumped at the top of every model step (begin_step): "this shadow was already cast in this step"


def begin_step():
    global STEP_ID
    STEP_ID += 1
    CTX.half_reports.clear()

WEIGHTS_EPOCH = 0          # bumped by anything that rewrites parameter storage behind autograd's back (optim.FusedClipAdamW
                           # updates the flat buffer through raw pointers: p._version and p.data_ptr() do not move)


def bump_weights_epoch():
    """Invalidate every bf16 weight shadow: the next forward re-casts from the fp32 masters."""
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1


def seed_word(device):
    """Device-resident u64 added to every dropout seed (see include/hriemo.h): a captured step bumps it
    in-graph so each replay draws fresh masks while kernel arguments stay frozen."""
    t = _seed_words.get(device.index)
    if t is None:
        t = torch.zeros(1, dtype=torch.int64, device=device)
        _seed_words[device.index] = t
    return t


DROP_LOG = None          # tests: a list that receives every dropout site of a forward in call order --
                         # ("attn", seed, site, B, H, Lq, Lk, p, b_off) / ("rows", seed, site, M, N, p, row_off); the effective seed
                         # is seed + the device seed word (tests/hashrng.py rebuilds the keep-masks from these)
_ones = {}


def one(device):
    """the constant 1.0 a training loop may hand to ``loss.backward(gradient=...)``: the fused losses recognise this very tensor
    and skip the multiplication by the incoming gradient (two elementwise launches and a fill on the step's serial chain)"""
    t = _ones.get(device.index)
    if t is None:
        t = _ones[device.index] = torch.ones((), dtype=torch.float32, device=device)
    return t


def _is_one(g):
    t = _ones.get(g.device.index) if g.is_cuda else None
    return t is not None and g.data_ptr() == t.data_ptr() and g.dim() == 0


def bump_seed_word(device):
    _lib.call("hriemo_seed_bump", _p(seed_word(device)), _stream())      # golden-ratio increment (mod 2^64)


def next_seed(training):
    """Per-call dropout seed drawn from torch's CPU generator (reproducible under torch.manual_seed;
    identical on every DP rank that seeded identically)."""
    if not training:
        return 0
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class Shadows:
    """bf16 copies of fp32 master weights, refreshed when the master changes (optimizer step,
    load_state_dict, .to())."""

    def __init__(self):
        self._d = {}

    def get(self, p):
        key = id(p)
        ent = self._d.get(key)
        ver = (p._version, p.data_ptr(), WEIGHTS_EPOCH)
        # inside a capture every step re-casts (the replayed graph must honour optimizer updates), but only once per step:
        # a shadow prefetched at the top of the step (prefetch()) is not cast again on the decoder's serial chain
        if (CTX.capturing and (ent is None or len(ent) < 3 or ent[2] != STEP_ID)) or ent is None or ent[0] != ver or ent[1].device != p.device:
            s = ent[1] if ent is not None and ent[1].device == p.device and ent[1].shape == p.shape else \
                torch.empty(p.shape, dtype=BF16, device=p.device)
            _require_gpu(p)
            _require_fp32_master(p)
            src = p.detach()
            if not src.is_contiguous():
                src = src.contiguous()
            _lib.call("hriemo_cast_f32_to_bf16", _p(src), _p(s), src.numel(), _stream())
            ent = (ver, s, STEP_ID)
            self._d[key] = ent
        return ent[1]

    def prefetch(self, params):
        for p in params:
            self.get(p)

    def stale(self, p):
        """would get(p) cast now?  -> (needs a cast, destination tensor, version key)"""
        ent = self._d.get(id(p))
        ver = (p._version, p.data_ptr(), WEIGHTS_EPOCH)
        need = (CTX.capturing and (ent is None or len(ent) < 3 or ent[2] != STEP_ID)) or ent is None or ent[0] != ver or ent[1].device != p.device
        dst = ent[1] if ent is not None and ent[1].device == p.device and ent[1].shape == p.shape else None
        return need, dst, ver

    def get_cat(self, parts):
        """bf16 shadow of the row-wise concatenation of master slices: parts = ((param, r0, r1), ...) -> [sum(r1 - r0), K].
        One GEMM per shared input (SharedProjFn) reads it; refreshed when any of the masters changes, like get()."""
        key = ("cat",) + tuple((id(p), r0, r1) for p, r0, r1 in parts)
        ent = self._d.get(key)
        ver = tuple((p._version, p.data_ptr()) for p, _, _ in parts) + (WEIGHTS_EPOCH,)
        dev = parts[0][0].device
        if (CTX.capturing and (ent is None or ent[2] != STEP_ID)) or ent is None or ent[0] != ver or ent[1].device != dev:
            rows = sum(r1 - r0 for _, r0, r1 in parts)
            K = parts[0][0].shape[1]
            s = ent[1] if ent is not None and ent[1].device == dev else torch.empty((rows, K), dtype=BF16, device=dev)
            at = 0
            for p, r0, r1 in parts:
                _require_gpu(p)
                _require_fp32_master(p)
                src = p.detach()
                if not src.is_contiguous():
                    src = src.contiguous()
                _lib.call("hriemo_cast_f32_to_bf16", _p(src[r0:r1]), _p(s[at:at + (r1 - r0)]), (r1 - r0) * K, _stream())
                at += r1 - r0
            ent = (ver, s, STEP_ID)
            self._d[key] = ent
        return ent[1]

    def get_cat_wb(self, wparts, bparts):
        """get_cat(wparts) and get_cat_vec(bparts) refreshed together by ONE launch (hriemo_cast_copy_batch) -- a shared
        projection's [3d, d] bf16 weight shadow and [3d] fp32 bias; before: two casts, a torch.cat and a copy per refresh."""
        kw = ("cat",) + tuple((id(p), r0, r1) for p, r0, r1 in wparts)
        kb = ("catv",) + tuple((id(p), r0, r1) for p, r0, r1 in bparts)
        ew, eb = self._d.get(kw), self._d.get(kb)
        vw = tuple((p._version, p.data_ptr()) for p, _, _ in wparts) + (WEIGHTS_EPOCH,)
        vb = tuple((p._version, p.data_ptr()) for p, _, _ in bparts) + (WEIGHTS_EPOCH,)
        dev = wparts[0][0].device
        stale_w = (CTX.capturing and (ew is None or ew[2] != STEP_ID)) or ew is None or ew[0] != vw or ew[1].device != dev
        stale_b = (CTX.capturing and (eb is None or eb[2] != STEP_ID)) or eb is None or eb[0] != vb or eb[1].device != dev
        if not (stale_w or stale_b):
            return ew[1], eb[1]
        if not all(p.is_contiguous() for p, _, _ in wparts + bparts):
            return self.get_cat(wparts), self.get_cat_vec(bparts)
        K = wparts[0][0].shape[1]
        rows = sum(r1 - r0 for _, r0, r1 in wparts)
        nb = sum(r1 - r0 for _, r0, r1 in bparts)
        sw = ew[1] if ew is not None and ew[1].device == dev and ew[1].shape == (rows, K) else torch.empty((rows, K), dtype=BF16, device=dev)
        sb = eb[1] if eb is not None and eb[1].device == dev and eb[1].shape == (nb,) else torch.empty((nb,), dtype=torch.float32, device=dev)
        jobs, at = [], 0
        for p, r0, r1 in wparts:
            _require_gpu(p)
            _require_fp32_master(p)
            jobs.append((p.data_ptr() + r0 * K * 4, sw.data_ptr() + at * K * 2, (r1 - r0) * K, 0))
            at += r1 - r0
        at = 0
        for p, r0, r1 in bparts:
            _require_fp32_master(p)
            jobs.append((p.data_ptr() + r0 * 4, sb.data_ptr() + at * 4, r1 - r0, 1))
            at += r1 - r0
        if any(j[0] % 16 or j[1] % 16 for j in jobs):
            return self.get_cat(wparts), self.get_cat_vec(bparts)
        host = torch.tensor(jobs, dtype=torch.int64)
        _lib.call("hriemo_cast_copy_batch", host.data_ptr(), len(jobs), _stream())
        self._d[kw] = (vw, sw, STEP_ID)
        self._d[kb] = (vb, sb, STEP_ID)
        return sw, sb

    def get_cat_mx8(self, parts):
        """MX-fp8 form (bytes [rows, K], scales [K/32, ld]) of the row-wise concatenation of master slices, refreshed like
        mx8_shadow: every slice is quantised from its fp32 master straight into its rows of ONE buffer (the scale matrix is
        k-block major with one column per row, so a slice owns a column range of it)"""
        key = ("catmx",) + tuple((id(p), r0, r1) for p, r0, r1 in parts)
        ent = self._d.get(key)
        ver = tuple((p._version, p.data_ptr()) for p, _, _ in parts) + (WEIGHTS_EPOCH,)
        if CTX.capturing or ent is None or ent[0] != ver:
            rows = sum(r1 - r0 for _, r0, r1 in parts)
            K = parts[0][0].shape[1]
            dev = parts[0][0].device
            L_ = _lib.lib()
            ld = L_.hriemo_mx8_scale_ld(rows)
            if ent is not None and ent[1][0].shape == (rows, K) and ent[1][0].device == dev:
                q, sc = ent[1]
            else:
                q = torch.empty((rows, K), dtype=torch.uint8, device=dev)
                sc = torch.zeros((K // 32, ld), dtype=torch.uint8, device=dev)
            at = 0
            for p, r0, r1 in parts:
                _require_gpu(p)
                _require_fp32_master(p)
                src = p.detach()
                if not src.is_contiguous():
                    src = src.contiguous()
                src = src[r0:r1]
                _lib.call("hriemo_quant_mx8", _p(src), src.stride(0), 1, r1 - r0, K, q.data_ptr() + at * K, K, sc.data_ptr() + at, ld,
                          _stream())
                at += r1 - r0
            ent = (ver, (q, sc))
            self._d[key] = ent
        return ent[1]

    def get_cat_vec(self, parts):
        """fp32 concatenation of bias slices ((param, r0, r1), ...), cached like the weight shadows"""
        key = ("catv",) + tuple((id(p), r0, r1) for p, r0, r1 in parts)
        ent = self._d.get(key)
        ver = tuple((p._version, p.data_ptr()) for p, _, _ in parts) + (WEIGHTS_EPOCH,)
        if (CTX.capturing and (ent is None or ent[2] != STEP_ID)) or ent is None or ent[0] != ver:
            v = torch.cat([p.detach()[r0:r1] for p, r0, r1 in parts])
            if ent is not None and ent[1].shape == v.shape and ent[1].device == v.device:
                ent[1].copy_(v)           # same storage: a captured graph keeps pointing at it
                v = ent[1]
            ent = (ver, v, STEP_ID)
            self._d[key] = ent
        return ent[1]


FUSED_WGRAD = True


def enable_fused_wgrad(params, on=True):
    """Opt the given parameters into in-place gradient accumulation (GradSink below).  dp.GradBuckets does this for
    the parameters whose .grad it owns.  The contract changes for them: the Functions return None for these gradients
    and write ``p.grad`` directly, bias / LayerNorm gradients are final only when backward() returns (launch-boundary
    reduce), ``torch.autograd.grad(loss, params)`` yields nothing for them, and tensor / post-accumulate hooks of other
    libraries (torch DDP, optimizer-in-backward) never see them -- do not combine with foreign gradient hooks."""
    for p in params:
        p._hriemo_fused_grad = bool(on)


class GradSink:
    """Where a Function's parameter gradients go.  Default: fresh tensors returned to autograd (hooks, autograd.grad and
    foreign reducers work as usual).  If every parameter of the sub-layer was opted in (enable_fused_wgrad; dp.GradBuckets
    does it for its flat-buffer views) and owns a dense fp32 ``.grad``, the kernels accumulate straight into it (GEMM /
    column-reduce ``accumulate`` flag) and the Function returns None for it: no temporary, no autograd add kernel."""

    def __init__(self, params):
        self.params = params
        self.fused = FUSED_WGRAD and all(
            getattr(p, "_hriemo_fused_grad", False)
            and p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous()
            and p.grad.device == p.device and p.grad.shape == p.shape for p in params)
        if self.fused:
            for p in params:
                if hasattr(p, "_hriemo_grad_ready"):
                    p._hriemo_sink_managed = True      # dp.GradBuckets: only done() below reports this gradient

    def buf(self, p):
        return p.grad if self.fused else torch.empty(p.shape, dtype=torch.float32, device=p.device)

    def ret(self, t):
        return None if self.fused else t

    def done(self, skip=(), after_flush=()):
        """skip: parameters another Function still adds to in this backward (it reports them); after_flush: matrices whose
        gradient is a column sum finished by the launch-boundary reduce, like the vectors"""
        if self.fused:
            for p in self.params:
                if any(p is q for q in skip):
                    continue
                hook = getattr(p, "_hriemo_grad_ready", None)
                if hook is not None:
                    if (p.dim() < 2 or any(p is q for q in after_flush)) and _small_dw.touches(p.grad):
                        _small_dw.add_hook(hook, p, True)     # its column sum is still queued (decoder / gate sized bias gradient)
                    elif (p.dim() < 2 or any(p is q for q in after_flush)) and DEFER_REDUCE and _in_backward():
                        _deferred.add_hook(hook, p)       # bias / LayerNorm gradients are final only after the flush
                    elif p.dim() >= 2 and _small_dw.touches(p.grad):
                        _small_dw.add_hook(hook, p)       # its weight-gradient GEMM is still queued
                    else:
                        hook(p)


def prefetch_batch(pairs):
    """bf16 shadows of many masters in ONE launch per 64 matrices (hriemo_cast_f32_to_bf16_batch): pairs = ((Shadows, param), ...).
    The gate's and the decoder's fourteen matrices at the top of a step: one ~25 us launch instead of a chain of fourteen
    dependent ~5 us ones in front of whatever runs next on that stream."""
    jobs, upd = [], []
    for sh, p in pairs:
        need, dst, ver = sh.stale(p)
        if not need:
            continue
        _require_gpu(p)
        _require_fp32_master(p)
        if not p.is_contiguous() or p.data_ptr() % 16:
            sh.get(p)
            continue
        if dst is None:
            dst = torch.empty(p.shape, dtype=BF16, device=p.device)
        jobs.append((p.data_ptr(), dst.data_ptr(), p.numel()))
        upd.append((sh, id(p), ver, dst))
    if jobs:
        host = torch.tensor(jobs, dtype=torch.int64)
        _lib.call("hriemo_cast_f32_to_bf16_batch", host.data_ptr(), len(jobs), _stream())
        for sh, key, ver, dst in upd:
            sh._d[key] = (ver, dst, STEP_ID)


def padded_shadow(sh, p, kp):
    """bf16 shadow of a [N,K] weight zero-padded to [N,kp] (kp = K rounded up to 8: 16-byte rows)"""
    key = (id(p), "pad")
    ent = sh._d.get(key)
    ver = (p._version, p.data_ptr(), WEIGHTS_EPOCH)
    if CTX.capturing or ent is None or ent[0] != ver or ent[1].device != p.device:
        _require_gpu(p)
        _require_fp32_master(p)
        s_ = ent[1] if ent is not None and ent[1].device == p.device else \
            torch.zeros((p.shape[0], kp), dtype=BF16, device=p.device)
        s_[:, :p.shape[1]].copy_(p.detach())
        ent = (ver, s_)
        sh._d[key] = ent
    return ent[1]


def to_bf16(x):
    return x if x.dtype == BF16 else x.to(BF16)


def mask_u8(mask, B, L):
    if mask is None:
        return None
    if mask.shape != (B, L):
        raise ValueError(f"key_padding_mask shape {tuple(mask.shape)} != {(B, L)}")
    m = mask.contiguous()
    return m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)


# ----------------------------------------------------------------------------- raw kernel wrappers
def gemm(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, c_f32=False, bias=None, epi=0, aux=None, ldaux=0, accumulate=False):
    ws = workspace(64 << 20, C.device) if c_f32 else None
    _lib.call("hriemo_gemm_bf16", ta, tb, M, N, K, _p(A), lda, _p(B), ldb, _p(C), ldc, int(c_f32), _p(bias), epi,
              _p(aux), ldaux, int(accumulate), _p(ws), ws.numel() * 4 if ws is not None else 0, _stream())


def linear_fwd(x, w16, bias, relu=False, out_f32=False):
    """x [M,K] bf16 (row stride free), w16 [N,K] bf16 -> [M,N]"""
    M, K = x.shape
    N = w16.shape[0]
    y = torch.empty((M, N), dtype=torch.float32 if out_f32 else BF16, device=x.device)
    gemm(0, 0, M, N, K, x, x.stride(0), w16, w16.stride(0), y, N, c_f32=out_f32, bias=bias, epi=1 if relu else 0)
    return y


def linear_dx(dy, w16, epi=0, aux=None):
    """dX[M,K] = dY[M,N] . W[N,K]  (optionally * (aux>0) or + aux)"""
    M, N = dy.shape
    K = w16.shape[1]
    dx = torch.empty((M, K), dtype=BF16, device=dy.device)
    gemm(0, 1, M, K, N, dy, dy.stride(0), w16, w16.stride(0), dx, K, epi=epi, aux=aux,
         ldaux=aux.stride(0) if aux is not None else 0)
    return dx


FOLD_FFN_BIAS = True       # the first FFN Linear's bias gradient out of the dX GEMM's epilogue (hriemo_gemm_bf16_colsum)


def fold_ffn_bias():
    return FOLD_FFN_BIAS


def linear_dx_masked_colsum(dy, w16, aux, bias_out, accumulate):
    """dX[M,K] = (dY . W) * (aux > 0) and bias_out[K] (+)= colsum(dX) in one pass: the GEMM's epilogue leaves per-row-block
    partial sums of the tile it stores, finished by the launch-boundary reduce inside backward (fused path) or right here"""
    M, N = dy.shape
    K = w16.shape[1]
    dx = torch.empty((M, K), dtype=BF16, device=dy.device)
    rows = _lib.lib().hriemo_gemm_colsum_rows(0, 1, M, K, N)
    part = torch.empty(rows * K, dtype=torch.float32, device=dy.device)
    _lib.call("hriemo_gemm_bf16_colsum", 0, 1, M, K, N, _p(dy), dy.stride(0), _p(w16), w16.stride(0), _p(dx), K, _p(aux), aux.stride(0),
              _p(part), _stream())
    if accumulate and DEFER_REDUCE and _in_backward():
        _deferred.add(part, K, rows, K, 1, [bias_out], True)
    else:
        red = _DeferredReduce()
        red.device = dy.device
        red.add(part, K, rows, K, 1, [bias_out], accumulate, schedule=False)
        red.flush()
    return dx


def linear_dw(dy, x, out, accumulate=False):
    """out[N,K] (fp32, row stride free) (+)= dY[M,N]^T . X[M,K]"""
    M, N = dy.shape
    K = x.shape[1]
    if accumulate and M <= SMALL_DW_ROWS and _small_dw.enabled():
        _small_dw.add(dy, x, out)          # decoder / gate sized: issued later, beside the encoder's backward (class below)
        return
    gemm(1, 1, N, K, M, dy, dy.stride(0), x, x.stride(0), out, out.stride(0), c_f32=True, accumulate=accumulate)


def linear_dw_split(dy, x, out, out2, split, accumulate=False):
    """[out ; out2] (+)= dY[M,N]^T . X[M,K]: rows [0, split) of the [N, K] result to `out`, the rest to `out2` (one launch)"""
    M, N = dy.shape
    K = x.shape[1]
    ws = workspace(64 << 20, out.device)
    _lib.call("hriemo_gemm_bf16_split", 1, 1, N, K, M, _p(dy), dy.stride(0), _p(x), x.stride(0), _p(out), out.stride(0), _p(out2),
              out2.stride(0), split, int(accumulate), _p(ws), ws.numel() * 4, _stream())


# ---- MX-fp8 operand path (forward projection / FFN GEMMs; include/hriemo.h "MX-fp8 operand path")
GEMM_MODE = None           # 'bf16' | 'mx_fp8'; read from HRIEMO_GEMM at first use, set_gemm_mode() changes it at run time


def set_gemm_mode(mode):
    """'bf16' (default) or 'mx_fp8': forward nn.Linear / in-proj / out-proj GEMMs on block-scaled fp8 operands (BASELINE.json
    configs[4]); attention cores, backward GEMMs, masters and gradients are unchanged."""
    global GEMM_MODE
    if mode not in ("bf16", "mx_fp8"):
        raise ValueError(f"gemm mode {mode!r}: expected 'bf16' or 'mx_fp8'")
    GEMM_MODE = mode


def gemm_mode():
    global GEMM_MODE
    if GEMM_MODE is None:
        import os
        GEMM_MODE = os.environ.get("HRIEMO_GEMM", "bf16")
        if GEMM_MODE not in ("bf16", "mx_fp8"):
            raise ValueError(f"HRIEMO_GEMM={GEMM_MODE!r}: expected 'bf16' or 'mx_fp8'")
    return GEMM_MODE


# ---- arithmetic of the whole path: 'bf16' (product path: bf16 GEMM operands, fp32 accumulate, fp32 residual twins) or 'fp32'
# (hri-emo_amd/_fp32.py: the reference's fp32 arithmetic to 1e-3, forward and backward).  HRIEMO_PRECISION at
# first use, set_precision().
PRECISION = None


def set_precision(mode):
    global PRECISION
    if mode not in ("bf16", "fp32"):
        raise ValueError(f"set_precision: {mode!r} (expected 'bf16' or 'fp32')")
    PRECISION = mode


def precision():
    global PRECISION
    if PRECISION is None:
        import os
        mode = os.environ.get("HRIEMO_PRECISION", "bf16")
        if mode not in ("bf16", "fp32"):
            raise ValueError(f"HRIEMO_PRECISION={mode!r} (expected 'bf16' or 'fp32')")
        PRECISION = mode
    return PRECISION


def _fp32():
    from . import _fp32 as m
    return m


MX_MIN_ROWS = 1024     # below this a GEMM is latency-bound (decoder queries, gate MLP): fp8 operands buy nothing there


def mx8_ok(K, N, M):
    """shapes the scaled-MFMA path takes (a 128-deep step per MFMA, enough rows to be throughput-bound); anything else stays on
    the bf16 GEMM"""
    return K % 128 == 0 and N % 8 == 0 and M >= MX_MIN_ROWS


def want_mx_copy(M, d):
    """should a LayerNorm also emit the MX-fp8 copy of its [M, d] output (it feeds projection / FFN GEMMs)?"""
    return gemm_mode() == "mx_fp8" and d % 128 == 0 and M >= MX_MIN_ROWS


def quant_mx8(x):
    """x [M,K] bf16 or fp32 (row stride free) -> (e4m3 bytes [M,K], E8M0 scales [K/32, ld])"""
    M, K = x.shape
    L_ = _lib.lib()
    ld = L_.hriemo_mx8_scale_ld(M)
    q = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    sc = torch.empty((K // 32, ld), dtype=torch.uint8, device=x.device)
    _lib.call("hriemo_quant_mx8", _p(x), x.stride(0), int(x.dtype == torch.float32), M, K, _p(q), K, _p(sc), ld, _stream())
    return q, sc


def linear_fwd_mx8(xq, xs, wq, ws, bias, relu=False, out_f32=False, want_q=False):
    """(xq, xs) [M,K] and (wq, ws) [N,K] quantised operands -> y [M,N] = x . w^T + bias; want_q: the epilogue also leaves the
    MX-fp8 form of y (the next GEMM's operand) attached to y (tag_mx): no quantisation pass over [M, N]"""
    M, K = xq.shape
    N = wq.shape[0]
    y = torch.empty((M, N), dtype=torch.float32 if out_f32 else BF16, device=xq.device)
    if want_q and not out_f32 and N % 128 == 0:
        ld = _lib.lib().hriemo_mx8_scale_ld(M)
        q = torch.empty((M, N), dtype=torch.uint8, device=xq.device)
        sc = torch.empty((N // 32, ld), dtype=torch.uint8, device=xq.device)
        _lib.call("hriemo_gemm_mx8_q", M, N, K, _p(xq), xq.stride(0), _p(xs), xs.stride(0), _p(wq), wq.stride(0), _p(ws), ws.stride(0),
                  _p(y), N, _p(bias), 1 if relu else 0, _p(q), N, _p(sc), ld, _stream())
        return tag_mx(y, (q, sc))
    _lib.call("hriemo_gemm_mx8", M, N, K, _p(xq), xq.stride(0), _p(xs), xs.stride(0), _p(wq), wq.stride(0), _p(ws), ws.stride(0),
              _p(y), N, int(out_f32), _p(bias), 1 if relu else 0, None, 0, _stream())
    return y


def mx8_shadow(sh, p, rows=None):
    """MX-fp8 copy (bytes, scales) of an fp32 master weight [N,K] (or of its row slice `rows`), refreshed like the bf16 shadow"""
    key = (id(p), "mx8", rows)
    ent = sh._d.get(key)
    ver = (p._version, p.data_ptr(), WEIGHTS_EPOCH)
    if CTX.capturing or ent is None or ent[0] != ver:
        _require_gpu(p)
        _require_fp32_master(p)
        src = p.detach()
        if rows is not None:
            src = src[rows[0]:rows[1]]
        if not src.is_contiguous():
            src = src.contiguous()
        ent = (ver, quant_mx8(src))
        sh._d[key] = ent
    return ent[1]


class Operand:
    """A GEMM input in the formats the forward may want: the bf16 rows and, lazily and at most once, their MX-fp8 form
    (handed over by the producing LayerNorm when it emitted one)."""
    __slots__ = ("x", "_q")

    def __init__(self, x, q=None):
        self.x, self._q = x, q

    def q(self):
        if self._q is None:
            self._q = quant_mx8(self.x)
        return self._q


def mx_of(t):
    """the MX-fp8 copy a producing LayerNorm attached to its output tensor (None if there is none)"""
    return getattr(t, "_hriemo_mx", None)


def tag_mx(t, mx):
    if mx is not None:
        t._hriemo_mx = mx
    return t


def proj_fwd(xop, sh, w, w16, bias, rows=None, relu=False, out_f32=False, want_q=False):
    """y = x . W[rows]^T + bias[rows] on the configured operand format.  xop: Operand or bf16 tensor; want_q (fp8 mode): y is the
    operand of another GEMM -- its MX-fp8 form comes out of this GEMM's epilogue, attached to y"""
    x = xop.x if isinstance(xop, Operand) else xop
    K = x.shape[1]
    N = (rows[1] - rows[0]) if rows is not None else w.shape[0]
    b = bias if rows is None or bias is None else bias[rows[0]:rows[1]]
    if gemm_mode() == "mx_fp8" and mx8_ok(K, N, x.shape[0]):
        xq, xs = xop.q() if isinstance(xop, Operand) else quant_mx8(x)
        wq, ws = mx8_shadow(sh, w, rows)
        return linear_fwd_mx8(xq, xs, wq, ws, b, relu=relu, out_f32=out_f32, want_q=want_q)
    wv = w16 if rows is None else w16[rows[0]:rows[1]]
    return linear_fwd(x, wv, b, relu=relu, out_f32=out_f32)


_GEMM_CONTENDED = None


def gemm_contended(on):
    """Will other kernels (RCCL collectives) hold CUs while the GEMMs run?  True: the loader / consumer GEMM draws its tiles from
    the per-XCD work queue (with 32 CUs held a 25600x3072x768 launch takes 140 us, 184 on the static walk; 118-124 alone); False
    (default): static walk, 5-8 % faster per launch on a chip the kernel has to itself, the same step time
    (profiles/r04_gemm_ws.log).  Bit 3 of hriemo_gemm_debug_flags.  No-op without a GPU."""
    global _GEMM_CONTENDED
    if _GEMM_CONTENDED is on or not torch.cuda.is_available():
        return
    L = _lib.lib()
    prev = L.hriemo_gemm_debug_flags(9)
    L.hriemo_gemm_debug_flags((prev & ~8) if on else (prev | 8))
    _GEMM_CONTENDED = on


FUSE_LN = None             # Linear + bias + dropout + residual + LayerNorm in one launch (csrc/gemm_ln.hip); HRIEMO_FUSE_LN=1
FUSE_LN_MIN_ROWS = 1024    # a full-row tile is one workgroup per 64 rows: the decoder's M = 384 would use 6 CUs


def fuse_ln(M, d):
    """should this sub-layer's output projection run fused with its LayerNorm?"""
    global FUSE_LN
    if FUSE_LN is None:
        import os
        FUSE_LN = os.environ.get("HRIEMO_FUSE_LN", "0") == "1"
    return FUSE_LN and gemm_mode() == "bf16" and M >= FUSE_LN_MIN_ROWS and bool(_lib.lib().hriemo_gemm_ln_supported(d))


def proj_add_ln_fwd(a, w16, bias, x, x32, gamma, beta, p, seed, site, row_off, want32, rows=None):
    """g = a . W^T + bias (bf16, kept for the backward); y = LN(x + drop(g)) -> (g, y, y32 | None, mean, rstd): the results of
    proj_fwd + add_ln_fwd from ONE kernel (full-row tiles, row statistics reduced across the waves of a workgroup)"""
    M, K = a.shape
    d = w16.shape[0]
    dev = a.device
    g = torch.empty((M, d), dtype=BF16, device=dev)
    y = torch.empty((M, d), dtype=BF16, device=dev)
    y32 = torch.empty((M, d), dtype=torch.float32, device=dev) if want32 else None
    mean = torch.empty(M, dtype=torch.float32, device=dev)
    rstd = torch.empty(M, dtype=torch.float32, device=dev)
    if DROP_LOG is not None and p > 0 and rows is None:
        DROP_LOG.append(("rows", seed, site, M, d, float(p), row_off))
    _lib.call("hriemo_gemm_ln_fwd", M, d, K, _p(a), a.stride(0), _p(w16), w16.stride(0), _p(bias), _p(x), _p(x32), _p(gamma), _p(beta),
              _p(g), _p(y), _p(y32), _p(mean), _p(rstd), _EPS, float(p), seed, _p(seed_word(dev)), site, row_off, _p(rows), _stream())
    return g, y, y32, mean, rstd


def colsum(x, out, accumulate=False):
    M, N = x.shape
    L_ = _lib.lib()
    if accumulate and M <= SMALL_DW_ROWS and not _small_dw.flushing and _small_dw.enabled():
        _small_dw.add(x, None, out)        # decoder / gate sized bias gradient: off the serial chain, like the weight gradients
        return
    if accumulate and DEFER_REDUCE and _in_backward() and not _small_dw.in_final:
        rows = L_.hriemo_colsum_partial_rows(M, N)
        part = torch.empty(rows * N, dtype=torch.float32, device=x.device)
        _lib.call("hriemo_colsum_bf16", _p(x), x.stride(0), M, N, None, 0, _p(part), _stream())
        _deferred.add(part, N, rows, N, 1, [out], True)
        return
    ws = workspace(L_.hriemo_colsum_workspace_bytes(M, N), x.device, slot=1)
    _lib.call("hriemo_colsum_bf16", _p(x), x.stride(0), M, N, _p(out), int(accumulate), _p(ws), _stream())


def _in_backward():
    """True while the autograd engine is running a backward pass on this thread (final callbacks can be queued)."""
    try:
        return torch._C._current_graph_task_id() != -1
    except AttributeError:
        return False


# Dropout keep-mask as bit words from the forward to the backward (include/hriemo.h, drop_mask_bits).  Measured at cfg 2: in the
# two-kernel backward the bit words save the hash but cost three registers in the dK/dV kernel (a wave per SIMD at head_dim 96)
# and a 2-byte store per lane and key tile in the forward -- no gain; in the single-pass backward (16 < L_k <= 128) they pay
# (profiles/r02_attention.log).  So a site asks for them exactly when its backward is the single kernel.
def attn_mask_bits(B, H, Lk, hd, Lq=None):
    if Lq is not None:
        return bool(_lib.lib().hriemo_attn_bwd_single_pass_q(B, H, Lq, Lk, hd))
    return bool(_lib.lib().hriemo_attn_bwd_single_pass(B, H, Lk, hd))


def attn_fwd(q, k, v, B, H, Lq, Lk, hd, kpm, p, seed, site, b_off, want_bits=False, cu=None):
    """-> (o, lse) or, with want_bits, (o, lse, mask_bits|None): the dropout keep-mask as bit words for the backward.
    cu = (cu_seqlens_q, cu_seqlens_k) int32 device tensors: q / k / v hold packed rows, Lq / Lk are the longest sequences"""
    o = torch.empty((q.shape[0], H * hd), dtype=BF16, device=q.device)
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=q.device)
    if DROP_LOG is not None and p > 0:
        DROP_LOG.append(("attn", seed, site, B, H, Lq, Lk, float(p), b_off))
    mb = None
    if want_bits and p > 0:
        mb = torch.empty(_lib.lib().hriemo_attn_mask_bytes(B, H, Lq, Lk) // 8, dtype=torch.int64, device=q.device)
    if cu is None and hd % 32 == 0 and want_mx_copy(q.shape[0], H * hd):
        # fp8 GEMM mode: the out-projection's operand leaves the attention kernel already quantised (tagged onto o)
        ld = _lib.lib().hriemo_mx8_scale_ld(q.shape[0])
        oq = torch.empty((q.shape[0], H * hd), dtype=torch.uint8, device=q.device)
        so = torch.empty((H * hd // 32, ld), dtype=torch.uint8, device=q.device)
        _lib.call("hriemo_attn_fwd_q", _p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0),
                  _p(kpm), _p(lse), B, H, Lq, Lk, hd, float(p), seed, _p(seed_word(q.device)), site, b_off, _p(mb), _p(oq), H * hd,
                  _p(so), ld, _stream())
        tag_mx(o, (oq, so))
        return (o, lse, mb) if want_bits else (o, lse)
    if cu is not None:
        if kpm is not None:
            raise ValueError("attn_fwd: packed sequences carry their lengths; no key_padding_mask")
        _lib.call("hriemo_attn_fwd_varlen", _p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0),
                  _p(cu[0]), _p(cu[1]), _p(lse), B, H, Lq, Lk, hd, float(p), seed, _p(seed_word(q.device)), site, b_off, _p(mb),
                  _stream())
    else:
        _lib.call("hriemo_attn_fwd", _p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0),
                  _p(kpm), _p(lse), B, H, Lq, Lk, hd, float(p), seed, _p(seed_word(q.device)), site, b_off, _p(mb), _stream())
    return (o, lse, mb) if want_bits else (o, lse)


def attn_bwd(q, k, v, o, do, dq, dk, dv, lse, B, H, Lq, Lk, hd, kpm, p, seed, site, b_off, bias_grad=None, mask_bits=None, cu=None):
    """bias_grad = (db_q [d], db_kv [2d]) fp32 views the column sums of dQ and dK|dV go to: the kernels leave per-block partial
    sums behind (fp32 values before the bf16 rounding of dQ/dK/dV); inside backward with the fused path they are finished by the
    launch-boundary reduce (accumulating), otherwise by one reduce launch right here (overwriting).  Returns True if it took
    care of them."""
    delta = torch.empty_like(lse)
    pq = pkv = None
    fold = bias_grad is not None and FOLD_ATTN_BIAS
    if fold:
        L_ = _lib.lib()
        rq, rk = L_.hriemo_attn_bwd_dq_colsum_rows(B, H, Lq, Lk, hd), L_.hriemo_attn_bwd_kv_colsum_rows(B, H, Lq, Lk, hd)
        pq = torch.empty(rq * H * hd, dtype=torch.float32, device=q.device)
        pkv = torch.empty(rk * 2 * H * hd, dtype=torch.float32, device=q.device)
    if cu is not None:
        _lib.call("hriemo_attn_bwd_varlen", _p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0),
                  _p(do), do.stride(0), _p(dq), dq.stride(0), _p(dk), dk.stride(0), _p(dv), dv.stride(0), _p(cu[0]), _p(cu[1]),
                  _p(lse), _p(delta), B, H, Lq, Lk, hd, float(p), seed, _p(seed_word(q.device)), site, b_off,
                  _p(pq), _p(pkv), _p(mask_bits), _stream())
    else:
        _lib.call("hriemo_attn_bwd", _p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0),
                  _p(do), do.stride(0), _p(dq), dq.stride(0), _p(dk), dk.stride(0), _p(dv), dv.stride(0), _p(kpm),
                  _p(lse), _p(delta), B, H, Lq, Lk, hd, float(p), seed, _p(seed_word(q.device)), site, b_off,
                  _p(pq), _p(pkv), _p(mask_bits), _stream())
    if fold:
        d = H * hd
        deferred = bias_grad[2] if len(bias_grad) > 2 else False
        if deferred and DEFER_REDUCE and _in_backward():
            _deferred.add(pq, d, rq, d, 1, [bias_grad[0]], True)
            _deferred.add(pkv, 2 * d, rk, 2 * d, 1, [bias_grad[1]], True)
        else:
            red = _DeferredReduce()
            red.device = q.device
            red.add(pq, d, rq, d, 1, [bias_grad[0]], deferred, schedule=False)
            red.add(pkv, 2 * d, rk, 2 * d, 1, [bias_grad[1]], deferred, schedule=False)
            red.flush()
    attn_bwd.last_delta = delta          # scratch of the last call (diagnostics: scripts_dev/dbg_attn.py)
    return fold


def attn_probs(q, k, B, H, Lq, Lk, hd, kpm, lse, p, seed, site, b_off):
    out = torch.empty((B, Lq, Lk), dtype=torch.float32, device=q.device)
    _lib.call("hriemo_attn_probs", _p(q), q.stride(0), _p(k), k.stride(0), _p(kpm), _p(lse), _p(out), B, H, Lq, Lk,
              hd, float(p), seed, _p(seed_word(q.device)), site, b_off, _stream())
    return out


def add_ln_fwd(g, x, gamma, beta, p, seed, site, row_off, x32=None, want32=False, want_mx=False, rows=None):
    """y = LN(x + drop(g)); x32 = fp32 twin of the residual stream (used instead of x when given);
    want32 -> also return the fp32 twin of y; want_mx -> also the MX-fp8 copy (bytes, scales) of y as a 5th result."""
    M, d = g.shape
    y = torch.empty((M, d), dtype=BF16, device=g.device)
    y32 = torch.empty((M, d), dtype=torch.float32, device=g.device) if want32 else None
    mean = torch.empty(M, dtype=torch.float32, device=g.device)
    rstd = torch.empty(M, dtype=torch.float32, device=g.device)
    if DROP_LOG is not None and p > 0 and rows is None:
        DROP_LOG.append(("rows", seed, site, M, d, float(p), row_off))
    if rows is not None:          # packed rows: the dropout hash is keyed by the row of the padded layout
        yq = ys = None
        ld = 0
        if want_mx:
            ld = _lib.lib().hriemo_mx8_scale_ld(M)
            yq = torch.empty((M, d), dtype=torch.uint8, device=g.device)
            ys = torch.empty((d // 32, ld), dtype=torch.uint8, device=g.device)
        _lib.call("hriemo_add_ln_fwd_rows", _p(g), _p(x), _p(x32), _p(gamma), _p(beta), _p(y), _p(y32), _p(mean), _p(rstd), M, d,
                  _EPS, float(p), seed, _p(seed_word(g.device)), site, row_off, _p(yq), _p(ys), ld, _p(rows), _stream())
        return (y, y32, mean, rstd, (yq, ys)) if want_mx else (y, y32, mean, rstd)
    if want_mx:
        ld = _lib.lib().hriemo_mx8_scale_ld(M)
        yq = torch.empty((M, d), dtype=torch.uint8, device=g.device)
        ys = torch.empty((d // 32, ld), dtype=torch.uint8, device=g.device)
        _lib.call("hriemo_add_ln_fwd_mx8", _p(g), _p(x), _p(x32), _p(gamma), _p(beta), _p(y), _p(y32), _p(mean), _p(rstd), M, d,
                  _EPS, float(p), seed, _p(seed_word(g.device)), site, row_off, _p(yq), _p(ys), ld, _stream())
        return y, y32, mean, rstd, (yq, ys)
    _lib.call("hriemo_add_ln_fwd", _p(g), _p(x), _p(x32), _p(gamma), _p(beta), _p(y), _p(y32), _p(mean), _p(rstd), M, d,
              _EPS, float(p), seed, _p(seed_word(g.device)), site, row_off, _stream())
    return y, y32, mean, rstd


def add_ln_bwd(dy, g, x, gamma, mean, rstd, p, seed, site, row_off, want_dx=True, outs=None, accumulate=False, x32=None, rows=None):
    """outs = (dgamma, dbeta, dbias) destination tensors (fp32 [d]); fresh ones when None.  rows: the row index that keys the
    dropout hash (packed sequences), see add_ln_fwd."""
    row_index = rows          # (`rows` below counts partial rows)
    M, d = g.shape
    dev = g.device
    dx = torch.empty((M, d), dtype=BF16, device=dev) if want_dx else None
    dg = torch.empty((M, d), dtype=BF16, device=dev) if (p > 0 or not want_dx) else None
    if outs is None:
        stats = torch.empty((3, d), dtype=torch.float32, device=dev)
        outs = (stats[0], stats[1], stats[2])
    L_ = _lib.lib()
    if accumulate and DEFER_REDUCE and _in_backward():
        rows = L_.hriemo_add_ln_bwd_partial_rows(M, d)
        part = torch.empty(rows * 3 * d, dtype=torch.float32, device=dev)
        _lib.call("hriemo_add_ln_bwd_rows", _p(dy), _p(g), _p(x), _p(x32), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dg), None,
                  None, None, 0, M, d, float(p), seed, _p(seed_word(dev)), site, row_off, _p(part), _p(row_index), _stream())
        _deferred.add(part, 3 * d, rows, d, 3 if outs[2] is not None else 2, [o for o in outs if o is not None], True)
    else:
        ws = workspace(L_.hriemo_add_ln_bwd_workspace_bytes(M, d), dev, slot=1)
        _lib.call("hriemo_add_ln_bwd_rows", _p(dy), _p(g), _p(x), _p(x32), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dg), _p(outs[0]),
                  _p(outs[1]), _p(outs[2]), int(accumulate), M, d, float(p), seed, _p(seed_word(dev)), site, row_off,
                  _p(ws), _p(row_index), _stream())
    if dg is None:
        dg = dx           # no dropout: both branches get the same gradient
    return dx, dg, outs[0], outs[1], outs[2]


TWIN = _os.environ.get("HRIEMO_FP32_TWIN", "1") != "0"      # carry the fp32 twin of the residual stream (LayerNorm outputs)


DEFER_REDUCE = True        # bias / LayerNorm column sums finished by ONE launch at the end of backward (_DeferredReduce)
FOLD_ATTN_BIAS = True      # in-proj bias grads from the attention backward kernels' own column sums


class _DeferredReduce:
    """Launch-boundary reduce (include/hriemo.h, hriemo_colreduce_batch): producers that accumulate straight into a
    parameter's .grad leave their per-block partial sums behind and ONE launch at the end of backward finishes all of
    them (85 five-microsecond launches per step otherwise).  The flush is an autograd-engine final callback: it runs
    on the caller's stream after the engine has joined every stream backward used."""

    def __init__(self):
        self.jobs, self.keep, self.scheduled, self.blocks, self.hooks = [], [], False, 0, []

    def _schedule(self):
        if not self.scheduled:
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)
            self.scheduled = True

    def add_hook(self, hook, p):
        self.hooks.append((hook, p))
        self._schedule()

    def add(self, part, pstride, np_, w, nseg, outs, accumulate, schedule=True):
        gx = (w + 31) // 32
        self.jobs.append([part.data_ptr(), pstride, np_, w, nseg | (int(bool(accumulate)) << 8) | (self.blocks << 32)]
                         + [o.data_ptr() for o in outs] + [0] * (3 - len(outs)))
        self.blocks += gx * nseg
        self.keep.append(part)
        self.keep.extend(outs)
        self.device = part.device
        if schedule:
            self._schedule()

    def flush(self):
        jobs, n, nblocks = self.jobs, len(self.jobs), self.blocks
        self.jobs, self.scheduled, self.blocks = [], False, 0
        if n:
            host = torch.tensor(jobs, dtype=torch.int64)
            dev = torch.empty((n, 8), dtype=torch.int64, device=self.device)
            _lib.call("hriemo_colreduce_batch", host.data_ptr(), n, _p(dev), nblocks, _stream())
            self.keep.append(dev)
        if not CTX.capturing and self.keep:     # a captured graph keeps using these buffers on every replay
            cur = torch.cuda.current_stream(self.device)
            for t in self.keep:     # partials of the text branch were allocated on the side stream
                t.record_stream(cur)
            self.keep = []
        hooks, self.hooks = self.hooks, []
        for hook, p in hooks:       # gradient-ready notifications held back until the values are final
            hook(p)


_deferred = _DeferredReduce()


# Deferred small weight-gradient GEMMs.  The decoder's and the gate's backward is a serial chain of latency-bound launches that
# runs before the encoder's backward starts; its 16 weight-gradient GEMMs (M = B*N_e = 384 or B rows of reduction) sit on that
# chain at ~20 us each although nothing downstream needs them.  When they accumulate straight into .grad (fused path) they are
# queued instead and issued in one go where the device has room: behind the LAST text-branch Function of backward (layer-0 text
# self-attention, on the side stream, while the audio branch still has ~0.5 ms of its own backward to run), or from an
# autograd-engine final callback if no such point comes.  Measured at cfg 2: 8.87 -> 8.7 ms per step (skipping them altogether:
# 8.65).  Off while gradient-ready hooks drive an overlapped exchange (the decoder's bucket would leave last instead of first);
# round 4: the same sites' bias-gradient column sums (five 5-7 us launches on the chain) are queued with them.
GATE_TWO_STREAMS = True    # the gate's text-side LayerNorm + pooling on the side stream
GATE_PAIR = True           # ... superseded for d <= 1024: both modalities' LayerNorm + pooling (forward and backward) from ONE launch
DEFER_SMALL_DW = True
GROUP_SMALL_DW = True      # the queued weight gradients leave as one grouped GEMM launch
SMALL_DW_ROWS = 1024
FLUSH_SITES = set()                # dropout-site ids of the sub-layers whose backward ends a model's text branch (one per CrossModalTransformer)
_hook_predicates = []              # one per live dp.GradBuckets with gradient-ready hooks (register_hook_predicate)


def register_hook_predicate(fn):
    """dp.GradBuckets: `fn()` is True while its gradient-ready hooks drive an overlapped exchange"""
    _hook_predicates.append(fn)
    return fn


def unregister_hook_predicate(fn):
    if fn in _hook_predicates:
        _hook_predicates.remove(fn)


def grad_hooks_active():
    return any(f() for f in _hook_predicates)


class _DeferredWgrad:
    def __init__(self):
        self.jobs, self.hooks, self.keep, self.scheduled = [], [], [], False
        self.flushing = self.in_final = False

    def enabled(self):
        return DEFER_SMALL_DW and _in_backward() and not grad_hooks_active()

    def add(self, dy, x, out):
        self.jobs.append((dy, x, out))
        if not self.scheduled:
            torch.autograd.Variable._execution_engine.queue_callback(self.final)
            self.scheduled = True

    def add_hook(self, hook, p, column_sum=False):
        self.hooks.append((hook, p, column_sum))

    def touches(self, g):
        if not self.jobs or g is None:
            return False
        a = g.data_ptr()
        b = a + g.numel() * g.element_size()
        return any(a <= j[2].data_ptr() < b for j in self.jobs)

    def flush(self):
        alljobs, self.jobs = self.jobs, []
        sums = [j for j in alljobs if j[1] is None]          # queued column sums (x is None): (dy, None, out)
        jobs = [j for j in alljobs if j[1] is not None]
        if sums:
            cur = torch.cuda.current_stream(sums[0][0].device)
            self.flushing = True
            try:
                for dy, _, out in sums:
                    colsum(dy, out, True)
                    if CTX.capturing:
                        self.keep.extend((dy, out))
                    else:
                        dy.record_stream(cur); out.record_stream(cur)
            finally:
                self.flushing = False
        if jobs:
            cur = torch.cuda.current_stream(jobs[0][0].device)
            grouped = GROUP_SMALL_DW and all(dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0 for dy, x, _ in jobs)
            if grouped:
                # ONE launch per 16 queued weight gradients (hriemo_gemm_bf16_group_tn) instead of one ~20 us launch each
                table = [(dy.shape[1], x.shape[1], dy.shape[0], dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), out.data_ptr(),
                          out.stride(0)) for dy, x, out in jobs]
                host = torch.tensor(table, dtype=torch.int64)
                _lib.call("hriemo_gemm_bf16_group_tn", host.data_ptr(), len(table), 1, _stream())
            for dy, x, out in jobs:
                M, N = dy.shape
                if not grouped:
                    gemm(1, 1, N, x.shape[1], M, dy, dy.stride(0), x, x.stride(0), out, out.stride(0), c_f32=True, accumulate=True)
                if CTX.capturing:
                    self.keep.extend((dy, x, out))      # a captured graph keeps reading these buffers on every replay
                else:
                    dy.record_stream(cur); x.record_stream(cur); out.record_stream(cur)
        hooks, self.hooks = self.hooks, []
        for hook, p, column_sum in hooks:
            if column_sum and DEFER_REDUCE and not self.in_final and _in_backward():
                _deferred.add_hook(hook, p)          # the column sum just issued is finished by the launch-boundary reduce
            else:
                hook(p)

    def final(self):
        self.scheduled = False
        self.in_final = True             # engine final callback: column sums finish in their own call, nothing is queued any more
        try:
            self.flush()
        finally:
            self.in_final = False


_small_dw = _DeferredWgrad()


def as_pair(x):
    """(bf16 copy for the GEMMs, fp32 twin for the residual path or None)"""
    if x.dtype == BF16:
        return x, None
    return x.to(BF16), (x if x.dtype == torch.float32 else x.float()) if TWIN else None


def from_pair(y16, y32, dtype):
    if dtype == BF16:
        return y16
    return (y32 if y32 is not None else y16).to(dtype)


def _sum_grads(d16, d32):
    """total upstream gradient of a (bf16, fp32-twin) output pair, as bf16"""
    if d32 is None:
        return d16
    if d16 is None:
        return d32.to(BF16)
    return (d16.float() + d32).to(BF16)


def _c32(t):
    return None if t is None else (t if t.is_contiguous() else t.contiguous())


def _heads(d, H):
    if d % H != 0:
        raise ValueError(f"d_model={d} not divisible by n_heads={H}")
    hd = d // H
    if hd not in (16, 32, 64, 96, 128):
        raise ValueError(f"head_dim={hd} is not built (supported: 16, 32, 64, 96, 128)")
    return hd


def _contig_bf16(t):
    t = to_bf16(t)
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------- packed (varlen) sequences, SURVEY 8(f) rank 4
# The reference pads every sample to the batch maximum and computes the PAD rows too (train_fusion_seq_level_decoder.py:191-232).
# Opt-in (HRIEMO_VARLEN=1 / set_varlen(True)): the cross-modal encoder -- 9/10 of the FLOPs -- runs on the VALID rows only: the
# row-wise kernels (GEMM, LayerNorm, FFN) see [N_valid, d] matrices, the attention kernels take cu_seqlens
# (hriemo_attn_*_varlen); the rows are scattered back to the padded layout before the gate, whose pooling, fusion and the
# decoder's masks only ever read valid positions -- so logits, beta and z are those of the padded path.  The packing needs the
# lengths on the host (one sync per distinct mask tensor); a step captured into a hipGraph bakes them in and may only be
# replayed with the same masks (dp.DataParallelStep checks).
VARLEN = None


def set_varlen(on):
    global VARLEN
    VARLEN = bool(on)


def varlen():
    global VARLEN
    if VARLEN is None:
        VARLEN = _os.environ.get("HRIEMO_VARLEN", "0") == "1"
    return VARLEN


class Seq:
    """packed rows of one modality: cu int32 [B+1] (device), idx int64 [N] packed row -> row of the padded [Breal*L] layout.
    Bucketed form (seq_bucket, one captured graph for every batch): N = the bucket's row count, the rows beyond the last real
    sequence form ONE extra sequence (B = Breal + 1, cu has B+1 entries) of zeros, so every kernel writes every row."""
    __slots__ = ("cu", "idx", "B", "L", "Lmax", "N", "Breal")

    def __init__(self, cu, idx, B, L, Lmax, N, Breal=None):
        self.cu, self.idx, self.B, self.L, self.Lmax, self.N = cu, idx, B, L, Lmax, N
        self.Breal = B if Breal is None else Breal


def seq_bucket(cu, B, L, n_rows):
    """Seq over `n_rows` packed rows whose lengths live in DEVICE memory: cu int32 [B+2] = [0, cumulative valid lengths of the B
    sequences ..., n_rows] (the caller refreshes it per batch; cu[B] < n_rows, n_rows - cu[B] <= L).  Nothing here depends on
    the lengths, so a step captured with it replays for every batch that fits the bucket."""
    idx = torch.empty(n_rows, dtype=torch.int64, device=cu.device)      # written by the first pack of the step (hriemo_pack_rows)
    return Seq(cu, idx, B + 1, L, L, n_rows, Breal=B)




_SEQ_PLANS = {}


def seq_plan(mask, B, L):
    """Seq for a [B, L] padding mask (True = PAD) whose valid positions are a prefix of every row with at least one valid
    position (what the collate produces); None when the mask does not have that form (the padded path then runs)."""
    if mask is None:
        return None
    key = (mask.data_ptr(), tuple(mask.shape), mask._version, str(mask.device))
    hit = _SEQ_PLANS.get(key)
    if hit is not None:
        return hit[0]
    if CTX.capturing:
        raise RuntimeError("varlen: the sequence lengths must be known before the step is captured (run one eager step first)")
    valid = ~mask.bool()
    lens = valid.sum(1)
    prefix = bool((valid == (torch.arange(L, device=mask.device)[None, :] < lens[:, None])).all()) and bool((lens > 0).all())
    plan = None
    if prefix:
        cu = torch.zeros(B + 1, dtype=torch.int32, device=mask.device)
        cu[1:] = torch.cumsum(lens, 0)
        idx = valid.reshape(-1).nonzero().reshape(-1)
        plan = Seq(cu, idx, B, L, int(lens.max()), int(idx.numel()))
    if len(_SEQ_PLANS) > 64:
        _SEQ_PLANS.clear()
    _SEQ_PLANS[key] = (plan, mask)            # the mask stays referenced: its address is the key
    return plan


class PackFn(torch.autograd.Function):
    """(x16 [B, L, d], x32 | None) -> packed ([1, N, d], twin | None): ONE gather launch for the pair (hriemo_pack_rows), bucket
    padding rows written as zeros; also fills seq.idx (padded row of every packed row)."""

    @staticmethod
    def forward(ctx, x16, x32, seq):
        ctx.set_materialize_grads(False)
        B, L, d = x16.shape
        x16c = _contig_bf16(x16)
        x32c = _c32(x32)
        p16 = torch.empty((1, seq.N, d), dtype=BF16, device=x16.device)
        p32 = torch.empty((1, seq.N, d), dtype=torch.float32, device=x16.device) if x32 is not None else None
        _lib.call("hriemo_pack_rows", _p(x16c), _p(x32c), _p(seq.cu), seq.Breal, L, d, seq.N, _p(p16), _p(p32), _p(seq.idx), _stream())
        ctx.seq, ctx.d, ctx.has32 = seq, d, x32 is not None
        return p16, p32

    @staticmethod
    def backward(ctx, d16, d32):
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) or (d16 is None and d32 is None):
            return None, None, None
        y16, y32 = _unpack_pair(d16, d32, ctx.seq, ctx.d)
        return y16, (y32 if ctx.has32 else None), None


def _unpack_pair(p16, p32, seq, d):
    dev = (p16 if p16 is not None else p32).device
    y16 = torch.empty((seq.Breal, seq.L, d), dtype=BF16, device=dev) if p16 is not None else None
    y32 = torch.empty((seq.Breal, seq.L, d), dtype=torch.float32, device=dev) if p32 is not None else None
    p16c = None if p16 is None else _contig_bf16(p16)
    _lib.call("hriemo_unpack_rows", _p(p16c), _p(_c32(p32)), _p(seq.cu), seq.Breal, seq.L, d, _p(y16), _p(y32), _stream())
    return y16, y32


class UnpackFn(torch.autograd.Function):
    """packed ([1, N, d], twin | None) -> padded ([B, L, d], twin | None) with zeros at the PAD positions: one scatter launch
    (hriemo_unpack_rows); backward = the gather of the pair's gradients (bucket padding rows get zeros)."""

    @staticmethod
    def forward(ctx, p16, p32, seq):
        ctx.set_materialize_grads(False)
        d = p16.shape[-1]
        ctx.seq, ctx.d = seq, d
        return _unpack_pair(p16, p32, seq, d)

    @staticmethod
    def backward(ctx, d16, d32):
        if d16 is None and d32 is None:
            return None, None, None
        seq, d = ctx.seq, ctx.d
        dev = (d16 if d16 is not None else d32).device
        g16 = torch.empty((1, seq.N, d), dtype=BF16, device=dev) if d16 is not None else None
        g32 = torch.empty((1, seq.N, d), dtype=torch.float32, device=dev) if d32 is not None else None
        d16c = None if d16 is None else _contig_bf16(d16)
        _lib.call("hriemo_pack_rows", _p(d16c), _p(_c32(d32)), _p(seq.cu), seq.Breal, seq.L, d, seq.N, _p(g16), _p(g32), None, _stream())
        return g16, g32, None


def pack_rows(x, seq):
    """[B, L, d] -> [1, N, d] by torch indexing: what PackFn computes, stated on seq.idx (host-logic tests; the model calls pack_pair)"""
    if x is None:
        return None
    B, L, d = x.shape
    return x.reshape(B * L, d).index_select(0, seq.idx).view(1, seq.N, d)


def unpack_rows(x, seq):
    """[1, N, d] -> [B, L, d] with zeros at the PAD positions by torch indexing (see pack_rows)"""
    if x is None:
        return None
    d = x.shape[-1]
    out = torch.zeros((seq.B * seq.L, d), dtype=x.dtype, device=x.device)
    return out.index_copy(0, seq.idx, x.reshape(seq.N, d)).view(seq.B, seq.L, d)


def pack_pair(x16, x32, seq):
    """[B, L, d] pair -> packed [1, N, d] pair (valid rows only)"""
    p16, p32 = PackFn.apply(x16, x32, seq)
    return p16, p32


def unpack_pair(p16, p32, seq):
    """packed [1, N, d] pair -> [B, L, d] pair with zeros at the PAD positions"""
    return UnpackFn.apply(p16, p32, seq)


# ----------------------------------------------------------------------------- sub-layer Functions
class _GradModeAware:
    """Function.forward always runs with grad mode off, and ctx.needs_input_grad mirrors the inputs' requires_grad even under
    torch.no_grad(): a forward that must know whether autograd is RECORDING this call (the fp32 mode saves its activations
    only then) reads the grad mode noted here at apply() time (thread-local)."""

    @classmethod
    def apply(cls, *args, **kwargs):
        prev = getattr(_apply_tls, "grad_mode", None)
        _apply_tls.grad_mode = torch.is_grad_enabled()
        try:
            return super().apply(*args, **kwargs)
        finally:
            _apply_tls.grad_mode = prev


def recording(ctx):
    """autograd will call this node's backward"""
    return bool(getattr(_apply_tls, "grad_mode", True)) and any(ctx.needs_input_grad)


class SelfAttnLN(_GradModeAware, torch.autograd.Function):
    """y = LN(x + drop(out_proj(MHA_core(in_proj(x))))) ; returns (y, probs|None)"""

    @staticmethod
    def forward(ctx, x, x32, w_in, b_in, w_out, b_out, gamma, beta, sh, H, kpm, p, seed, site, b_off, need_w):
        if precision() == "fp32":
            return _fp32().self_attn_ln(ctx, x, x32, w_in, b_in, w_out, b_out, gamma, beta, sh, H, kpm, need_w, p, seed, site, b_off)
        _require_fp32_masters(w_in, b_in, w_out, b_out, gamma, beta)
        ctx.set_materialize_grads(False)      # an unused twin output must arrive as None, not as zeros
        _require_gpu(x)
        B, L, d = x.shape
        hd = _heads(d, H)
        M = B * L
        # packed sequences (x is [1, N_valid, d], the kpm slot carries the Seq): the attention sees AB samples of up to AL rows
        AB, AL, cu, RL, rows = B, L, None, L, None          # RL / rows: row stride and row index that key the LayerNorm dropout
        if isinstance(kpm, Seq):
            if need_w:
                raise ValueError("attention maps are exported by the padded path only")
            AB, AL, cu, RL, rows, kpm = kpm.B, kpm.Lmax, (kpm.cu, kpm.cu), kpm.L, kpm.idx, None
        x2 = _contig_bf16(x).view(M, d)
        x32 = _c32(x32)
        x32v = x32.view(M, d) if x32 is not None else None
        w_in16, w_out16 = sh.get(w_in), sh.get(w_out)
        qkv = proj_fwd(Operand(x2, mx_of(x)), sh, w_in, w_in16, b_in)
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        o, lse, mbits = attn_fwd(q, k, v, AB, H, AL, AL, hd, kpm, p, seed, site, b_off, want_bits=True, cu=cu) if attn_mask_bits(AB, H, AL, hd, AL) else \
            attn_fwd(q, k, v, AB, H, AL, AL, hd, kpm, p, seed, site, b_off, cu=cu) + (None,)
        if fuse_ln(M, d):
            g, y, y32, mean, rstd = proj_add_ln_fwd(o, w_out16, b_out, x2, x32v, gamma, beta, p, seed, site + 1, b_off * RL, TWIN, rows)
            mx = []
        else:
            g = proj_fwd(Operand(o, mx_of(o)), sh, w_out, w_out16, b_out)
            y, y32, mean, rstd, *mx = add_ln_fwd(g, x2, gamma, beta, p, seed, site + 1, b_off * RL, x32=x32v, want32=TWIN,
                                                 want_mx=want_mx_copy(M, d), rows=rows)
        probs = attn_probs(q, k, B, H, L, L, hd, kpm, lse, p, seed, site, b_off) if need_w else None
        ctx.save_for_backward(x2, x32v, qkv, o, lse, g, mean, rstd, w_in16, w_out16, gamma, kpm, mbits)
        ctx.cfg = (B, L, d, H, hd, p, seed, site, b_off)
        ctx.packed = (AB, AL, cu, RL, rows)
        ctx.params = (w_in, b_in, w_out, b_out, gamma, beta)
        ctx.mark_non_differentiable(*( [probs] if probs is not None else []))
        return tag_mx(y.view(B, L, d), mx[0] if mx else None), (y32.view(B, L, d) if y32 is not None else None), probs

    @staticmethod
    def backward(ctx, dy, dy32, _dprobs):
        if getattr(ctx, "fp32", False):
            return _fp32().self_attn_ln_bwd(ctx, dy, dy32)
        x2, x32v, qkv, o, lse, g, mean, rstd, w_in16, w_out16, gamma, kpm, mbits = ctx.saved_tensors
        B, L, d, H, hd, p, seed, site, b_off = ctx.cfg
        AB, AL, cu, RL, rows = ctx.packed
        M = B * L
        dev = x2.device
        dy2 = _contig_bf16(_sum_grads(dy, dy32)).view(M, d)
        p_w_in, p_b_in, p_w_out, p_b_out, p_gamma, p_beta = ctx.params
        sink = GradSink(ctx.params)
        acc = sink.fused
        ds, dg, dgamma, dbeta, db_out = add_ln_bwd(dy2, g, x2, gamma, mean, rstd, p, seed, site + 1, b_off * RL,
                                                   outs=(sink.buf(p_gamma), sink.buf(p_beta), sink.buf(p_b_out)),
                                                   accumulate=acc, x32=x32v, rows=rows)
        dw_out = sink.buf(p_w_out)
        linear_dw(dg, o, dw_out, acc)
        do = linear_dx(dg, w_out16)
        dqkv = torch.empty((M, 3 * d), dtype=BF16, device=dev)
        db_in = sink.buf(p_b_in)
        folded = attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, do, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:],
                          lse, AB, H, AL, AL, hd, kpm, p, seed, site, b_off, bias_grad=(db_in[:d], db_in[d:], acc), mask_bits=mbits,
                          cu=cu)
        dw_in = sink.buf(p_w_in)
        linear_dw(dqkv, x2, dw_in, acc)
        if not folded:
            colsum(dqkv, db_in, acc)
        dx = linear_dx(dqkv, w_in16, epi=3, aux=ds)
        sink.done()
        if site in FLUSH_SITES:
            _small_dw.flush()                 # the text branch's backward ends here: queued decoder / gate weight gradients go out now
        r = sink.ret
        return (dx.view(B, L, d), None, r(dw_in), r(db_in), r(dw_out), r(db_out), r(dgamma), r(dbeta)) + (None,) * 8


class CrossAttnLN(_GradModeAware, torch.autograd.Function):
    """y = LN(xq + drop(out_proj(MHA_core(Wq xq, Wkv xkv)))) ; returns (y, probs|None)"""

    @staticmethod
    def forward(ctx, xq, xq32, xkv, w_in, b_in, w_out, b_out, gamma, beta, sh, H, kpm, p, seed, site, b_off, need_w, kv_pre=None,
                join_q=None, q_pre=None, slots=None):
        """kv_pre: the K | V projection of xkv computed by KVProjFn ahead of time (the decoder hoists it onto the side stream); its
        weight-gradient and dX then belong to that Function, this one returns dK | dV for it.
        q_pre: likewise the Q projection of xq (SharedProjFn: one GEMM per shared input); this Function then only returns dQ for it
        and hands the residual-path gradient of xq to join_q.  slots = (SharedGrad of dQ, SharedGrad of dK|dV): where the attention
        backward writes those gradients, so that they arrive at the projection's backward as column slices of ONE buffer."""
        if precision() == "fp32":
            return _fp32().cross_attn_ln(ctx, xq, xq32, xkv, w_in, b_in, w_out, b_out, gamma, beta, sh, H, kpm, need_w, p, seed, site, b_off)
        _require_fp32_masters(w_in, b_in, w_out, b_out, gamma, beta)
        ctx.set_materialize_grads(False)      # an unused twin output must arrive as None, not as zeros
        _require_gpu(xq)
        B, Lq, d = xq.shape
        Lk = xkv.shape[1]
        hd = _heads(d, H)
        # packed sequences: the kpm slot carries (Seq of the query side, Seq of the key side)
        AB, ALq, ALk, cu, RL, rows = B, Lq, Lk, None, Lq, None
        if isinstance(kpm, tuple):
            if need_w:
                raise ValueError("attention maps are exported by the padded path only")
            sq, sk = kpm
            AB, ALq, ALk, cu, RL, rows, kpm = sq.B, sq.Lmax, sk.Lmax, (sq.cu, sk.cu), sq.L, sq.idx, None
        xq2 = _contig_bf16(xq).view(B * Lq, d)
        xq32 = _c32(xq32)
        x32v = xq32.view(B * Lq, d) if xq32 is not None else None
        w_out16 = sh.get(w_out)
        w_in16 = sh.get(w_in) if (q_pre is None or kv_pre is None) else None
        q = q_pre if q_pre is not None else proj_fwd(Operand(xq2, mx_of(xq)), sh, w_in, w_in16, b_in, rows=(0, d))
        if kv_pre is not None:
            xkv2, kv = None, kv_pre
        else:
            xkv2 = _contig_bf16(xkv).view(B * Lk, d)
            kv = proj_fwd(Operand(xkv2, mx_of(xkv)), sh, w_in, w_in16, b_in, rows=(d, 3 * d))
        k, v = kv[:, :d], kv[:, d:]
        o, lse, mbits = attn_fwd(q, k, v, AB, H, ALq, ALk, hd, kpm, p, seed, site, b_off, want_bits=True, cu=cu) if attn_mask_bits(AB, H, ALk, hd, ALq) else \
            attn_fwd(q, k, v, AB, H, ALq, ALk, hd, kpm, p, seed, site, b_off, cu=cu) + (None,)
        if fuse_ln(B * Lq, d):
            g, y, y32, mean, rstd = proj_add_ln_fwd(o, w_out16, b_out, xq2, x32v, gamma, beta, p, seed, site + 1, b_off * RL, TWIN, rows)
            mx = []
        else:
            g = proj_fwd(Operand(o, mx_of(o)), sh, w_out, w_out16, b_out)
            y, y32, mean, rstd, *mx = add_ln_fwd(g, xq2, gamma, beta, p, seed, site + 1, b_off * RL, x32=x32v, want32=TWIN,
                                                 want_mx=want_mx_copy(B * Lq, d), rows=rows)
        probs = attn_probs(q, k, B, H, Lq, Lk, hd, kpm, lse, p, seed, site, b_off) if need_w else None
        ctx.save_for_backward(xq2, x32v, xkv2, q, kv, o, lse, g, mean, rstd, w_in16, w_out16, gamma, kpm, mbits)
        ctx.cfg = (B, Lq, Lk, d, H, hd, p, seed, site, b_off)
        ctx.packed = (AB, ALq, ALk, cu, RL, rows)
        ctx.kv_pre = kv_pre is not None
        ctx.q_pre = q_pre is not None
        ctx.slots = slots
        ctx.join_q = join_q
        ctx.params = (w_in, b_in, w_out, b_out, gamma, beta)
        ctx.mark_non_differentiable(*([probs] if probs is not None else []))
        return tag_mx(y.view(B, Lq, d), mx[0] if mx else None), (y32.view(B, Lq, d) if y32 is not None else None), probs

    @staticmethod
    def _backward_shared(ctx, dy, dy32):
        """q_pre and kv_pre: both projections belong to SharedProjFn nodes.  Left here: LayerNorm / out-projection backward, the
        attention core's backward (dQ, dK|dV written into the projections' shared gradient buffers, in-projection bias gradients
        from its column sums), and the residual-path gradient of xq, deposited for the projection's dX GEMM."""
        xq2, x32v, xkv2, q, kv, o, lse, g, mean, rstd, w_in16, w_out16, gamma, kpm, mbits = ctx.saved_tensors
        B, Lq, Lk, d, H, hd, p, seed, site, b_off = ctx.cfg
        AB, ALq, ALk, cu, RL, rows = ctx.packed
        dy2 = _contig_bf16(_sum_grads(dy, dy32)).view(B * Lq, d)
        p_w_in, p_b_in, p_w_out, p_b_out, p_gamma, p_beta = ctx.params
        sink = GradSink((p_b_in, p_w_out, p_b_out, p_gamma, p_beta))
        acc = sink.fused
        ds, dg, dgamma, dbeta, db_out = add_ln_bwd(dy2, g, xq2, gamma, mean, rstd, p, seed, site + 1, b_off * RL,
                                                   outs=(sink.buf(p_gamma), sink.buf(p_beta), sink.buf(p_b_out)),
                                                   accumulate=acc, x32=x32v, rows=rows)
        dw_out = sink.buf(p_w_out)
        linear_dw(dg, o, dw_out, acc)
        do = linear_dx(dg, w_out16)
        sq, skv = ctx.slots
        dq, dkv = sq.buf(), skv.buf()
        db_in = sink.buf(p_b_in)
        folded = attn_bwd(q, kv[:, :d], kv[:, d:], o, do, dq, dkv[:, :d], dkv[:, d:], lse, AB, H, ALq, ALk, hd, kpm, p, seed,
                          site, b_off, bias_grad=(db_in[:d], db_in[d:], acc), mask_bits=mbits, cu=cu)
        if not folded:
            colsum(dq, db_in[:d], acc)
            colsum(dkv, db_in[d:], acc)
        dxq = ds
        if ctx.join_q is not None:
            ctx.join_q.arrive()
            ctx.join_q.deposit(ds)            # the projection's dX GEMM adds it in its epilogue
            dxq = None
        sink.done()
        r = sink.ret
        return (dxq.view(B, Lq, d) if dxq is not None else None, None, None, None, r(db_in), r(dw_out), r(db_out), r(dgamma),
                r(dbeta)) + (None,) * 8 + (dkv, None, dq, None)

    @staticmethod
    def backward(ctx, dy, dy32, _dprobs):
        if getattr(ctx, "fp32", False):
            return _fp32().cross_attn_ln_bwd(ctx, dy, dy32)
        if ctx.q_pre and ctx.kv_pre and ctx.slots is not None:
            return CrossAttnLN._backward_shared(ctx, dy, dy32)
        xq2, x32v, xkv2, q, kv, o, lse, g, mean, rstd, w_in16, w_out16, gamma, kpm, mbits = ctx.saved_tensors
        B, Lq, Lk, d, H, hd, p, seed, site, b_off = ctx.cfg
        AB, ALq, ALk, cu, RL, rows = ctx.packed
        dev = xq2.device
        dy2 = _contig_bf16(_sum_grads(dy, dy32)).view(B * Lq, d)
        p_w_in, p_b_in, p_w_out, p_b_out, p_gamma, p_beta = ctx.params
        sink = GradSink(ctx.params)
        acc = sink.fused
        ds, dg, dgamma, dbeta, db_out = add_ln_bwd(dy2, g, xq2, gamma, mean, rstd, p, seed, site + 1, b_off * RL,
                                                   outs=(sink.buf(p_gamma), sink.buf(p_beta), sink.buf(p_b_out)),
                                                   accumulate=acc, x32=x32v, rows=rows)
        dw_out = sink.buf(p_w_out)
        linear_dw(dg, o, dw_out, acc)
        do = linear_dx(dg, w_out16)
        dq = torch.empty((B * Lq, d), dtype=BF16, device=dev)
        dkv = torch.empty((B * Lk, 2 * d), dtype=BF16, device=dev)
        db_in = sink.buf(p_b_in)
        folded = attn_bwd(q, kv[:, :d], kv[:, d:], o, do, dq, dkv[:, :d], dkv[:, d:], lse, AB, H, ALq, ALk, hd, kpm, p, seed,
                          site, b_off, bias_grad=(db_in[:d], db_in[d:], acc), mask_bits=mbits, cu=cu)
        dw_in = sink.buf(p_w_in)
        linear_dw(dq, xq2, dw_in[:d], acc)
        if not folded:
            colsum(dq, db_in[:d], acc)
            colsum(dkv, db_in[d:], acc)
        dxq = linear_dx(dq, w_in16[:d], epi=3, aux=ds)
        if ctx.join_q is not None:            # xq has another consumer (GradJoin): deposit for it, or finish the sum if it came first
            last, dep = ctx.join_q.arrive()
            if dep is not None:
                dxq = dxq + dep.view(B * Lq, d)          # this GEMM's aux slot carries the residual gradient: explicit add
            if not last:
                ctx.join_q.deposit(dxq)
                dxq = None
        dxq = dxq.view(B, Lq, d) if dxq is not None else None
        r = sink.ret
        if ctx.kv_pre:
            # the K | V projection belongs to KVProjFn: it gets dK | dV, adds its rows of dW_in and reports the parameter
            if not acc:
                dw_in[d:].zero_()
            sink.done(skip=(p_w_in,))
            return (dxq, None, None, r(dw_in), r(db_in), r(dw_out), r(db_out), r(dgamma), r(dbeta)) + (None,) * 8 + (dkv, None, None, None)
        linear_dw(dkv, xkv2, dw_in[d:], acc)
        dxkv = linear_dx(dkv, w_in16[d:])
        sink.done()
        return (dxq, None, dxkv.view(B, Lk, d), r(dw_in), r(db_in), r(dw_out), r(db_out), r(dgamma),
                r(dbeta)) + (None,) * 12


class GradJoin:
    """One tensor, n consumers that are Functions of this file (the audio self-attention output feeds the a2t queries AND the t2a
    keys / values; the decoder's memory feeds every layer): autograd would sum their gradients with an elementwise add launch per
    pair.  Here every consumer but the last DEPOSITS its gradient and returns None, the last folds the deposit into its dX
    GEMM's add-aux epilogue (epi = 3) and returns the total.  "Last" is decided by arrival, so any execution order of the
    consumers' backward is correct; a consumer that cannot fold (its GEMM's aux slot is taken) adds explicitly.  If a consumer's
    backward never runs, the deposit would be lost: an engine final callback checks that every join was completed."""

    def __init__(self, n=2):
        self.n, self.seen, self.dep, self.ev = n, 0, None, None

    def arrive(self):
        """-> (is_last, deposit | None): the deposit is ordered before the caller's stream on return"""
        if self.seen == 0:
            torch.autograd.Variable._execution_engine.queue_callback(self._check)
        self.seen += 1
        last = self.seen == self.n
        dep, ev = self.dep, self.ev
        self.dep = self.ev = None
        if last:
            self.seen = 0
        if dep is not None:
            cur = torch.cuda.current_stream(dep.device)
            if ev != cur:                      # `ev` is the depositor's stream
                cur.wait_stream(ev)
                dep.record_stream(cur)         # also inside a capture: the depositor's stream must not reuse it under the reader
        return last, dep

    def deposit(self, t):
        self.dep = t
        self.ev = torch.cuda.current_stream(t.device)

    def _check(self):
        if self.seen != 0 or self.dep is not None:
            self.seen, self.dep, self.ev = 0, None, None
            raise RuntimeError("GradJoin: a consumer of a shared activation did not run its backward; its partner's gradient was "
                               "deposited for it (_ops.GRAD_JOIN = False lets autograd sum the gradients)")


GRAD_JOIN = True           # (False: autograd sums the gradients of a shared activation -- what the tests hold the joins against)
SHARED_PROJ = True         # one N = 3d projection GEMM per shared encoder input (SharedProjFn); False: Q and K | V apart


def shared_proj():
    return SHARED_PROJ




def grad_join(n=2, always=False):
    """a GradJoin where every consumer is certain to run its backward: the encoder's joins only inside the fusion model's forward
    (a stand-alone CrossModalBlock may be trained on one of its two outputs, and the unused branch's consumers never run), the
    decoder's (its layers are a chain) always"""
    return GradJoin(n) if (GRAD_JOIN and torch.is_grad_enabled() and (always or CTX.join_scope > 0)) else None


class SharedGrad:
    """[M, 3d] gradient buffer of one SharedProjFn output, filled by two attention backward launches (dQ by the cross-attention
    this input queries, dK | dV by the one it serves as keys / values -- different streams), so that the projection's backward
    reads ONE dY operand.  A slot is a column range of it; the buffer is allocated by whichever slot is asked first."""

    def __init__(self, M, d, device):
        # Allocated HERE, in the forward, on the stream of its projection and before the branches exchange their K | V halves
        # (a point both streams are ordered behind): never in backward by whichever attention core comes first.  A buffer born
        # in backward on one stream can be written by the OTHER stream before the allocating stream has run the kernels that
        # still read the block's previous occupant -- the allocator only orders reuse within the allocating stream.  That was a
        # real race: the text branch's dQ landed in a block the audio branch's pending FFN dX GEMM still read its residual
        # operand from (scripts_dev/soak_step.py 500 16 ragged: a handful of dirty replays in 500, first differing word always
        # layer 1's attn_a2t.out_proj gradient).  Cost: (M_a + M_t) x 3d bf16 per layer held from forward to backward (312 MB
        # at cfg 2).
        self.M, self.d, self.device = M, d, device
        self.t = torch.empty((M, 3 * d), dtype=BF16, device=device) if torch.is_grad_enabled() else None
        self.stream = torch.cuda.current_stream(device) if self.t is not None else None

    def slot(self, c0, c1):
        return _SharedSlot(self, c0, c1)

    def guard(self, cur):
        """the buffer is used on `cur`, possibly not the stream it was allocated on: the allocator must not hand its memory to a
        later allocation of the allocating stream while `cur` still works on it.  Also inside a capture (there the block simply
        stays reserved until the capture ends): without it a replayed step raced -- the determinism test caught it."""
        if self.t is not None and self.stream != cur:
            self.t.record_stream(cur)

    def take(self):
        self.guard(torch.cuda.current_stream(self.device))          # the projection's backward reads it on its own stream
        t, self.t = self.t, None
        return t


class _SharedSlot:
    def __init__(self, owner, c0, c1):
        self.owner, self.c0, self.c1 = owner, c0, c1

    def buf(self):
        o = self.owner
        cur = torch.cuda.current_stream(o.device)
        if o.t is None:                       # (forward ran under no_grad, or the buffer was already consumed: a second backward)
            o.t = torch.empty((o.M, 3 * o.d), dtype=BF16, device=o.device)
            o.stream = cur
            o.late = True
        else:
            o.guard(cur)
            if getattr(o, "late", False) and o.stream != cur:
                cur.wait_stream(o.stream)     # born in backward after all: order this stream behind the allocating one first
        return o.t[:, self.c0:self.c1]


def _report_half(p, sink):
    """an in-projection weight's gradient is written by TWO SharedProjFn nodes (rows [:d] by the projection of the tensor it
    queries, rows [d:] by the one it serves as keys / values): the gradient-ready notification goes out with the second"""
    hook = getattr(p, "_hriemo_grad_ready", None)
    if not sink.fused or hook is None:
        return
    n = CTX.half_reports.get(id(p), 0) + 1
    if n == 2:
        CTX.half_reports.pop(id(p), None)
        hook(p)
    else:
        CTX.half_reports[id(p)] = n


class SharedProjFn(torch.autograd.Function):
    """[q | kv][M, 3d] = x . [W_q ; W_kv]^T + [b_q ; b_kv] -- ONE N = 3d GEMM per shared input (reference call
    sites models/cross_modal_block_tacfn.py:98-104,111-117): each self-attention output is the query input of its own
    cross-attention (rows [:d] of that module's in_proj_weight) and the key / value input of the other one (rows [d:] of the
    other module's).  Forward: one GEMM on the row-concatenated bf16 shadow; backward: one K = 3d dX GEMM (the residual-path
    gradient of x from the cross-attention's LayerNorm arrives through `join` and is added in the epilogue) and the weight
    gradients from the same [M, 3d] dY buffer.  state_dict is untouched: the parameters stay where the reference has them."""

    @staticmethod
    def forward(ctx, x, wq, bq, wkv, bkv, sh, join, shared):
        _require_fp32_masters(wq, bq, wkv, bkv)
        _require_gpu(x)
        ctx.set_materialize_grads(False)      # a half nobody differentiated must arrive as None, not as zeros
        B, L, d = x.shape
        x2 = _contig_bf16(x).view(B * L, d)
        wcat, bcat = sh.get_cat_wb(((wq, 0, d), (wkv, d, 3 * d)), ((bq, 0, d), (bkv, d, 3 * d)))
        if gemm_mode() == "mx_fp8" and mx8_ok(d, 3 * d, B * L):
            # cfg 5: the same ONE N = 3d launch on MX-fp8 operands (the LayerNorm that produced x left its fp8 copy; the
            # concatenated weight is quantised slice by slice into one buffer); the backward below is the bf16 one either way
            xq, xs = Operand(x2, mx_of(x)).q()
            wq8, ws8 = sh.get_cat_mx8(((wq, 0, d), (wkv, d, 3 * d)))
            out = linear_fwd_mx8(xq, xs, wq8, ws8, bcat)
        else:
            out = linear_fwd(x2, wcat, bcat)
        ctx.save_for_backward(x2, wcat)
        ctx.cfg = (B, L, d)
        ctx.join, ctx.shared = join, shared
        ctx.params = (wq, wkv)
        return out[:, :d], out[:, d:]

    @staticmethod
    def backward(ctx, dq, dkv):
        x2, wcat = ctx.saved_tensors
        B, L, d = ctx.cfg
        M = B * L
        dcat = ctx.shared.take()
        if dq is None and dkv is None:
            return (None,) * 8
        ok = (dcat is not None and dq is not None and dkv is not None and dq.data_ptr() == dcat.data_ptr()
              and dkv.data_ptr() == dcat.data_ptr() + 2 * d and dq.stride(0) == 3 * d and dkv.stride(0) == 3 * d)
        if not ok:                                # gradients that did not come through the shared buffer (or only one of them)
            dcat = torch.zeros((M, 3 * d), dtype=BF16, device=x2.device)
            if dq is not None:
                dcat[:, :d].copy_(dq)
            if dkv is not None:
                dcat[:, d:].copy_(dkv)
        wq, wkv = ctx.params
        sq, skv = GradSink((wq,)), GradSink((wkv,))
        dwq = dwkv = None                         # a half whose gradient never arrived leaves its weight's .grad alone (None, like
        if dq is not None:                        # autograd would for a cross-attention that took no part in the loss)
            dwq = sq.buf(wq)
            if not sq.fused:
                dwq[d:].zero_()
        if dkv is not None:
            dwkv = skv.buf(wkv)
            if not skv.fused:
                dwkv[:d].zero_()
        if dwq is not None and dwkv is not None and sq.fused == skv.fused:
            linear_dw_split(dcat, x2, dwq[:d], dwkv[d:], d, sq.fused)        # one 3d x d x M weight-gradient GEMM, two destinations
        else:
            if dwq is not None:
                linear_dw(dcat[:, :d], x2, dwq[:d], sq.fused)
            if dwkv is not None:
                linear_dw(dcat[:, d:], x2, dwkv[d:], skv.fused)
        dep = None
        if ctx.join is not None:
            _, dep = ctx.join.arrive()
        dx = linear_dx(dcat, wcat, epi=3, aux=dep.view(M, d)) if dep is not None else linear_dx(dcat, wcat)
        _report_half(wq, sq)
        _report_half(wkv, skv)
        return dx.view(B, L, d), (sq.ret(dwq) if dwq is not None else None), None, (skv.ret(dwkv) if dwkv is not None else None), None, None, None, None


class KVProjFn(torch.autograd.Function):
    """kv[B*L_k, 2d] = x_kv . W_in[d:3d]^T + b_in[d:3d]: the key / value half of a cross-attention's packed in-projection as a node
    of its own, so that the decoder can run it (forward and backward) on the side stream beside its serial chain of
    latency-bound launches -- the memory does not depend on the queries (models/emotion_decoder.py:48-54).  The bias gradient
    stays with CrossAttnLN (it comes out of the attention backward's column sums)."""

    @staticmethod
    def forward(ctx, xkv, w_in, b_in, sh, join=None):
        _require_fp32_masters(w_in, b_in)
        _require_gpu(xkv)
        ctx.join = join
        B, Lk, d = xkv.shape
        xkv2 = _contig_bf16(xkv).view(B * Lk, d)
        w_in16 = sh.get(w_in)
        kv = proj_fwd(Operand(xkv2, mx_of(xkv)), sh, w_in, w_in16, b_in, rows=(d, 3 * d))
        ctx.save_for_backward(xkv2, w_in16)
        ctx.cfg = (B, Lk, d)
        ctx.params = (w_in,)
        return kv

    @staticmethod
    def backward(ctx, dkv):
        xkv2, w_in16 = ctx.saved_tensors
        B, Lk, d = ctx.cfg
        dkv = _contig_bf16(dkv)
        sink = GradSink(ctx.params)
        acc = sink.fused
        dw_in = sink.buf(ctx.params[0])
        if not acc:
            dw_in[:d].zero_()
        linear_dw(dkv, xkv2, dw_in[d:], acc)
        last, dep = ctx.join.arrive() if ctx.join is not None else (True, None)
        dxkv = linear_dx(dkv, w_in16[d:], epi=3, aux=dep.view(B * Lk, d)) if dep is not None else linear_dx(dkv, w_in16[d:])
        sink.done()
        if not last:
            ctx.join.deposit(dxkv)
            return None, sink.ret(dw_in), None, None, None
        return dxkv.view(B, Lk, d), sink.ret(dw_in), None, None, None


class FFNLN(_GradModeAware, torch.autograd.Function):
    """y = LN(x + drop(W2 . drop_mid(relu(W1 x + b1)) + b2))"""

    @staticmethod
    def forward(ctx, x, x32, w1, b1, w2, b2, gamma, beta, sh, p, p_mid, seed, site, b_off, seq=None):
        """seq: the Seq of packed rows (x is [1, N_valid, d]): keys the LayerNorm dropout by the rows of the padded layout"""
        if precision() == "fp32":
            return _fp32().ffn_ln(ctx, x, x32, w1, b1, w2, b2, gamma, beta, sh, p, p_mid, seed, site, b_off)
        _require_fp32_masters(w1, b1, w2, b2, gamma, beta)
        ctx.set_materialize_grads(False)      # an unused twin output must arrive as None, not as zeros
        _require_gpu(x)
        B, L, d = x.shape
        M = B * L
        x2 = _contig_bf16(x).view(M, d)
        x32 = _c32(x32)
        x32v = x32.view(M, d) if x32 is not None else None
        w1_16, w2_16 = sh.get(w1), sh.get(w2)
        h = proj_fwd(Operand(x2, mx_of(x)), sh, w1, w1_16, b1, relu=True, want_q=p_mid == 0)
        hd_ = h
        if p_mid > 0:
            hd_ = torch.empty_like(h)
            if DROP_LOG is not None:
                DROP_LOG.append(("rows", seed, site + 2, M, h.shape[1], float(p_mid), b_off * L))
            _lib.call("hriemo_dropout_bf16", _p(h), _p(hd_), M, h.shape[1], float(p_mid), seed, _p(seed_word(h.device)),
                      site + 2, b_off * L, _stream())
        RL, rows = (seq.L, seq.idx) if seq is not None else (L, None)
        if seq is not None and p_mid > 0:
            raise ValueError("FFNLN: packed rows with a mid-FFN dropout are not built (the encoder's FFNs have none)")
        ctx.rowkey = (RL, rows)
        if fuse_ln(M, d):
            g, y, y32, mean, rstd = proj_add_ln_fwd(hd_, w2_16, b2, x2, x32v, gamma, beta, p, seed, site + 1, b_off * RL, TWIN, rows)
            mx = []
        else:
            g = proj_fwd(Operand(hd_, mx_of(hd_)), sh, w2, w2_16, b2)
            y, y32, mean, rstd, *mx = add_ln_fwd(g, x2, gamma, beta, p, seed, site + 1, b_off * RL, rows=rows, x32=x32v, want32=TWIN,
                                                 want_mx=want_mx_copy(M, d))
        ctx.save_for_backward(x2, x32v, h, hd_, g, mean, rstd, w1_16, w2_16, gamma)
        ctx.cfg = (B, L, d, p, p_mid, seed, site, b_off)
        ctx.params = (w1, b1, w2, b2, gamma, beta)
        return tag_mx(y.view(B, L, d), mx[0] if mx else None), (y32.view(B, L, d) if y32 is not None else None)

    @staticmethod
    def backward(ctx, dy, dy32):
        if getattr(ctx, "fp32", False):
            return _fp32().ffn_ln_bwd(ctx, dy, dy32)
        x2, x32v, h, hd_, g, mean, rstd, w1_16, w2_16, gamma = ctx.saved_tensors
        B, L, d, p, p_mid, seed, site, b_off = ctx.cfg
        M, F = h.shape
        dev = x2.device
        dy2 = _contig_bf16(_sum_grads(dy, dy32)).view(M, d)
        p_w1, p_b1, p_w2, p_b2, p_gamma, p_beta = ctx.params
        sink = GradSink(ctx.params)
        acc = sink.fused
        RL, rows = ctx.rowkey
        ds, dg, dgamma, dbeta, db2 = add_ln_bwd(dy2, g, x2, gamma, mean, rstd, p, seed, site + 1, b_off * RL,
                                                outs=(sink.buf(p_gamma), sink.buf(p_beta), sink.buf(p_b2)),
                                                accumulate=acc, x32=x32v, rows=rows)
        dw2 = sink.buf(p_w2)
        linear_dw(dg, hd_, dw2, acc)
        db1 = sink.buf(p_b1)
        fold = p_mid == 0 and fold_ffn_bias()
        if fold:                                         # db1 = colsum(da) out of the GEMM that writes da
            da = linear_dx_masked_colsum(dg, w2_16, h, db1, acc)
        else:
            da = linear_dx(dg, w2_16, epi=2, aux=h)          # * relu'(h)
        if p_mid > 0:
            _lib.call("hriemo_dropout_bf16", _p(da), _p(da), M, F, float(p_mid), seed, _p(seed_word(dev)), site + 2, b_off * L,
                      _stream())
        dw1 = sink.buf(p_w1)
        linear_dw(da, x2, dw1, acc)
        if not fold:
            colsum(da, db1, acc)
        dx = linear_dx(da, w1_16, epi=3, aux=ds)
        sink.done()
        r = sink.ret
        return (dx.view(B, L, d), None, r(dw1), r(db1), r(dw2), r(db2), r(dgamma), r(dbeta)) + (None,) * 7


class BetaGateFn(_GradModeAware, torch.autograd.Function):
    """(h_fusion, beta) = BetaGate(h_a, h_t, masks)  -- models/beta_gate_tacfn.py:68-118"""

    @staticmethod
    def forward(ctx, h_a, h_a32, h_t, h_t32, ga, ba, gt, bt, w1, b1, w2, b2, sh, kpm_a, kpm_t):
        if precision() == "fp32":            # -> (h_fusion as the fp32 tensor itself, beta)
            return _fp32().beta_gate(ctx, h_a, h_a32, h_t, h_t32, ga, ba, gt, bt, w1, b1, w2, b2, sh, kpm_a, kpm_t)
        _require_fp32_masters(ga, ba, gt, bt, w1, b1, w2, b2)
        _require_gpu(h_a)
        h_a32, h_t32 = _c32(h_a32), _c32(h_t32)
        B, La, d = h_a.shape
        Lt = h_t.shape[1]
        L = La if La == Lt else Lt                      # :98-104 (align to the text length)
        if La < L:
            raise RuntimeError(f"BetaGate: audio length {La} < text length {Lt}; the reference cannot fuse this either")
        dev = h_a.device
        xa, xt = _contig_bf16(h_a), _contig_bf16(h_t)
        f32 = dict(dtype=torch.float32, device=dev)
        L_ = _lib.lib()
        nca, nct = L_.hriemo_pool_chunks(La), L_.hriemo_pool_chunks(Lt)
        An = torch.empty((B, L, d), dtype=BF16, device=dev)
        Tn = torch.empty((B, L, d), dtype=BF16, device=dev)
        mean_a, rstd_a = torch.empty(B * La, **f32), torch.empty(B * La, **f32)
        mean_t, rstd_t = torch.empty(B * Lt, **f32), torch.empty(B * Lt, **f32)
        pa, pt = torch.empty((B, nca, d), **f32), torch.empty((B, nct, d), **f32)
        st = _stream()
        # the two modalities' LayerNorm + pooling are independent: the (small) text one runs on the side stream beside the audio one
        main = torch.cuda.current_stream(dev)
        side = side_stream(dev) if GATE_TWO_STREAMS else None
        if GATE_PAIR and L_.hriemo_ln_pool_pair_supported(d):
            # both modalities from one launch: no fork / join around the step (two graph edges and 10-27 us of idle device each)
            _lib.call("hriemo_ln_pool_fwd_pair", _p(xa), _p(h_a32), _p(kpm_a), _p(ga), _p(ba), _p(An), _p(mean_a), _p(rstd_a), _p(pa), La,
                      _p(xt), _p(h_t32), _p(kpm_t), _p(gt), _p(bt), _p(Tn), _p(mean_t), _p(rstd_t), _p(pt), Lt, B, L, d, _EPS, st)
        elif side is not None and side != main:
            fork(side, main)
            with torch.cuda.stream(side):
                _lib.call("hriemo_ln_pool_fwd", _p(xt), _p(h_t32), _p(kpm_t), _p(gt), _p(bt), _p(Tn), _p(mean_t), _p(rstd_t), _p(pt),
                          B, Lt, L, d, _EPS, _stream())
            _lib.call("hriemo_ln_pool_fwd", _p(xa), _p(h_a32), _p(kpm_a), _p(ga), _p(ba), _p(An), _p(mean_a), _p(rstd_a), _p(pa),
                      B, La, L, d, _EPS, st)
            main.wait_stream(side)
            if not CTX.capturing:
                for t_ in (xt, h_t32, kpm_t, Tn, mean_t, rstd_t, pt):
                    share(t_, side)
        else:
            _lib.call("hriemo_ln_pool_fwd", _p(xa), _p(h_a32), _p(kpm_a), _p(ga), _p(ba), _p(An), _p(mean_a), _p(rstd_a), _p(pa),
                      B, La, L, d, _EPS, st)
            _lib.call("hriemo_ln_pool_fwd", _p(xt), _p(h_t32), _p(kpm_t), _p(gt), _p(bt), _p(Tn), _p(mean_t), _p(rstd_t), _p(pt),
                      B, Lt, L, d, _EPS, st)
        gin = torch.empty((B, 4 * d), dtype=BF16, device=dev)
        a_pool, t_pool = torch.empty((B, d), **f32), torch.empty((B, d), **f32)
        cnt = torch.empty((B, 2), **f32)
        _lib.call("hriemo_gate_input", _p(pa), _p(pt), _p(kpm_a), _p(kpm_t), B, La, Lt, d, _p(gin), _p(a_pool),
                  _p(t_pool), _p(cnt), st)
        w1_16, w2_16 = sh.get(w1), sh.get(w2)
        hid = linear_fwd(gin, w1_16, b1, relu=True)
        pre = linear_fwd(hid, w2_16, b2, out_f32=True)
        w = torch.empty((B, d), **f32)
        beta = torch.empty((B, 1), **f32)
        _lib.call("hriemo_sigmoid_beta", _p(pre), _p(w), _p(beta), B, d, st)
        H = torch.empty((B, L, d), dtype=BF16, device=dev)
        _lib.call("hriemo_fuse_fwd", _p(w), _p(An), _p(Tn), _p(H), B, L, d, st)
        ctx.save_for_backward(xa, xt, An, Tn, mean_a, rstd_a, mean_t, rstd_t, gin, a_pool, t_pool, cnt, hid, w, w1_16,
                              w2_16, ga, gt, kpm_a, kpm_t, h_a32, h_t32)
        ctx.cfg = (B, La, Lt, L, d)
        ctx.params = (ga, ba, gt, bt, w1, b1, w2, b2)
        return H, beta

    @staticmethod
    def backward(ctx, dH, dbeta):
        if getattr(ctx, "fp32", False):
            return _fp32().beta_gate_bwd(ctx, dH, dbeta)
        (xa, xt, An, Tn, mean_a, rstd_a, mean_t, rstd_t, gin, a_pool, t_pool, cnt, hid, w, w1_16, w2_16, ga, gt, kpm_a,
         kpm_t, h_a32, h_t32) = ctx.saved_tensors
        B, La, Lt, L, d = ctx.cfg
        dev = xa.device
        f32 = dict(dtype=torch.float32, device=dev)
        st = _stream()
        L_ = _lib.lib()
        p_ga, p_ba, p_gt, p_bt, p_w1, p_b1, p_w2, p_b2 = ctx.params
        # parameter gradients go straight into .grad when the parameters were opted in (dp.GradBuckets): no autograd accumulate
        # launches on the serial chain between the decoder's and the encoder's backward
        sink = GradSink(ctx.params)
        acc = sink.fused
        dH2 = _contig_bf16(dH) if dH is not None else torch.zeros((B, L, d), dtype=BF16, device=dev)
        dbeta2 = dbeta.contiguous().float() if dbeta is not None else None
        part = torch.empty((B, L_.hriemo_pool_chunks(L), d), **f32)
        _lib.call("hriemo_fuse_bwd_dw", _p(dH2), _p(An), _p(Tn), _p(part), B, L, d, st)
        dpre = torch.empty((B, d), dtype=BF16, device=dev)
        _lib.call("hriemo_gate_dpre", _p(part), L, _p(dbeta2), _p(w), _p(dpre), B, d, st)
        Hd = hid.shape[1]
        dw2 = sink.buf(p_w2)
        linear_dw(dpre, hid, dw2, acc)
        db2 = sink.buf(p_b2)
        colsum(dpre, db2, acc)
        dhid = linear_dx(dpre, w2_16, epi=2, aux=hid)
        dw1 = sink.buf(p_w1)
        linear_dw(dhid, gin, dw1, acc)
        db1 = sink.buf(p_b1)
        colsum(dhid, db1, acc)
        dgin = linear_dx(dhid, w1_16)
        da, dt = torch.empty((B, d), **f32), torch.empty((B, d), **f32)
        _lib.call("hriemo_gate_input_bwd", _p(dgin), _p(a_pool), _p(t_pool), _p(cnt), _p(da), _p(dt), B, d, st)
        dxa = torch.empty((B, La, d), dtype=BF16, device=dev)
        dxt = torch.empty((B, Lt, d), dtype=BF16, device=dev)
        dga, dba, dgt, dbt = sink.buf(p_ga), sink.buf(p_ba), sink.buf(p_gt), sink.buf(p_bt)
        # LayerNorm-affine gradients: finished by the launch-boundary reduce when they accumulate into .grad inside backward (the
        # kernel's partial sums then live in a buffer of their own instead of the shared workspace), else by the call's own reduce
        defer = acc and DEFER_REDUCE and _in_backward()
        nba, nbt = L_.hriemo_ln_pool_bwd_workspace_bytes(B, La, d), L_.hriemo_ln_pool_bwd_workspace_bytes(B, Lt, d)

        def ln_pool_bwd(is_a, dpool, kpm, x, x32, gamma, mean, rstd, dx, dgam, dbet, Lx, nbytes):
            if defer:
                wsx = torch.empty(nbytes // 4 + 16, **f32)
                _lib.call("hriemo_ln_pool_bwd", _p(dH2), L, _p(w), is_a, _p(dpool), _p(kpm), _p(x), _p(x32), _p(gamma), _p(mean),
                          _p(rstd), _p(dx), None, None, 1, B, Lx, d, _p(wsx), _stream())
                _deferred.add(wsx, 2 * d, B * L_.hriemo_ln_pool_bwd_chunks(Lx), d, 2, [dgam, dbet], True)
            else:
                wsx = workspace(nbytes, dev, slot=1)
                _lib.call("hriemo_ln_pool_bwd", _p(dH2), L, _p(w), is_a, _p(dpool), _p(kpm), _p(x), _p(x32), _p(gamma), _p(mean),
                          _p(rstd), _p(dx), _p(dgam), _p(dbet), int(acc), B, Lx, d, _p(wsx), _stream())

        main = torch.cuda.current_stream(dev)
        side = side_stream(dev) if GATE_TWO_STREAMS else None
        if GATE_PAIR and L_.hriemo_ln_pool_pair_supported(d):
            nca_, nct_ = L_.hriemo_ln_pool_bwd_chunks(La), L_.hriemo_ln_pool_bwd_chunks(Lt)
            if defer:
                wsa, wst = torch.empty(nba // 4 + 16, **f32), torch.empty(nbt // 4 + 16, **f32)
                outs = (None,) * 4
            else:
                wsa = workspace(nba + nbt + 256, dev, slot=1)
                wst = wsa[(nba // 4 + 63) // 64 * 64:]
                outs = (dga, dba, dgt, dbt)
            _lib.call("hriemo_ln_pool_bwd_pair", _p(dH2), L, _p(w),
                      _p(da), _p(kpm_a), _p(xa), _p(h_a32), _p(ga), _p(mean_a), _p(rstd_a), _p(dxa), _p(outs[0]), _p(outs[1]), La, _p(wsa),
                      _p(dt), _p(kpm_t), _p(xt), _p(h_t32), _p(gt), _p(mean_t), _p(rstd_t), _p(dxt), _p(outs[2]), _p(outs[3]), Lt, _p(wst),
                      int(acc), B, d, _stream())
            if defer:
                _deferred.add(wsa, 2 * d, B * nca_, d, 2, [dga, dba], True)
                _deferred.add(wst, 2 * d, B * nct_, d, 2, [dgt, dbt], True)
        elif side is not None and side != main:
            # text on the side stream (its own workspace there), audio on this one; joined before the gradients are handed back
            fork(side, main)
            with torch.cuda.stream(side):
                ln_pool_bwd(0, dt, kpm_t, xt, h_t32, gt, mean_t, rstd_t, dxt, dgt, dbt, Lt, nbt)
            ln_pool_bwd(1, da, kpm_a, xa, h_a32, ga, mean_a, rstd_a, dxa, dga, dba, La, nba)
            main.wait_stream(side)
            if not CTX.capturing:
                for t_ in (dH2, w, dt, dxt, dgt, dbt):
                    share(t_, side)
        else:
            ln_pool_bwd(1, da, kpm_a, xa, h_a32, ga, mean_a, rstd_a, dxa, dga, dba, La, nba)
            ln_pool_bwd(0, dt, kpm_t, xt, h_t32, gt, mean_t, rstd_t, dxt, dgt, dbt, Lt, nbt)
        sink.done()
        r = sink.ret
        return dxa, None, dxt, None, r(dga), r(dba), r(dgt), r(dbt), r(dw1), r(db1), r(dw2), r(db2), None, None, None


class LegacyBetaGateFn(torch.autograd.Function):
    """(h_fusion, beta) of the legacy scalar gate, models/beta_gate.py:60-114: masked-mean pools of the raw features,
    [a, t, |a-t|, a*t] -> Linear(4d,h) -> ReLU -> Linear(h,1) -> sigmoid = beta[B,1], h = beta*h_a[:, :L] + (1-beta)*h_t[:, :L].
    Everything runs in libhriemo.so: pooling / gate input / fusion row kernels, the MLP on hriemo_gemm_bf16 (first layer) and
    hriemo_rowdot_* (the h -> 1 layer), like the vector gate; only [B]- and [B,h]-sized casts / masks are torch plumbing."""

    @staticmethod
    def forward(ctx, h_a, h_t, w1, b1, w2, b2, sh, kpm_a, kpm_t):
        if precision() == "fp32":
            raise NotImplementedError("HRIEMO_PRECISION=fp32 covers the tacfn model family (FusionWithEmotionDecoder); the legacy "
                                      "scalar gate (models/beta_gate.py) runs on the bf16 path only")
        _require_fp32_masters(w1, b1, w2, b2)
        _require_gpu(h_a)
        B, La, d = h_a.shape
        Lt = h_t.shape[1]
        L = La if La == Lt else Lt                      # beta_gate.py:97-101
        if La < L:
            raise RuntimeError(f"BetaGate: audio length {La} < text length {Lt}; the reference cannot fuse this either")
        dev = h_a.device
        xa, xt = _contig_bf16(h_a), _contig_bf16(h_t)
        f32 = dict(dtype=torch.float32, device=dev)
        st = _stream()
        a_pool, t_pool = torch.empty((B, d), **f32), torch.empty((B, d), **f32)
        cnt_a, cnt_t = torch.empty(B, **f32), torch.empty(B, **f32)
        _lib.call("hriemo_masked_mean_fwd", _p(xa), _p(kpm_a), _p(a_pool), _p(cnt_a), B, La, d, st)
        _lib.call("hriemo_masked_mean_fwd", _p(xt), _p(kpm_t), _p(t_pool), _p(cnt_t), B, Lt, d, st)
        gin = torch.empty((B, 4 * d), dtype=BF16, device=dev)
        _lib.call("hriemo_gate_input_pooled", _p(a_pool), _p(t_pool), _p(gin), B, d, st)
        w1_16 = sh.get(w1)
        hid = linear_fwd(gin, w1_16, b1, relu=True)                                        # [B, h] bf16
        w2f, b2f = w2.detach().float().contiguous(), b2.detach().float().contiguous()
        pre = torch.empty(B, **f32)
        _lib.call("hriemo_rowdot_fwd", _p(hid), None, _p(w2f), _p(b2f), _p(pre), B, hid.shape[1], st)
        beta = torch.empty((B, 1), **f32)
        wfull1 = torch.empty((B, 1), **f32)
        _lib.call("hriemo_sigmoid_beta", _p(pre), _p(wfull1), _p(beta), B, 1, st)          # d = 1: w == beta == sigmoid(pre)
        A = xa if La == L else xa[:, :L].contiguous()
        T = xt
        wfull = beta.expand(B, d).contiguous()
        H = torch.empty((B, L, d), dtype=BF16, device=dev)
        _lib.call("hriemo_fuse_fwd", _p(wfull), _p(A), _p(T), _p(H), B, L, d, st)
        ctx.save_for_backward(A, T, cnt_a, cnt_t, kpm_a, kpm_t, a_pool, t_pool, gin, hid, beta, w1_16, w2f)
        ctx.cfg = (B, La, Lt, L, d)
        return H, beta

    @staticmethod
    def backward(ctx, dH, dbeta):
        A, T, cnt_a, cnt_t, kpm_a, kpm_t, a_pool, t_pool, gin, hid, beta, w1_16, w2f = ctx.saved_tensors
        B, La, Lt, L, d = ctx.cfg
        dev = A.device
        f32 = dict(dtype=torch.float32, device=dev)
        st = _stream()
        L_ = _lib.lib()
        dH2 = _contig_bf16(dH) if dH is not None else torch.zeros((B, L, d), dtype=BF16, device=dev)
        nc = L_.hriemo_pool_chunks(L)
        part = torch.empty((B, nc, d), **f32)
        _lib.call("hriemo_fuse_bwd_dw", _p(dH2), _p(A), _p(T), _p(part), B, L, d, st)
        dbeta_h = torch.empty(B, **f32)
        _lib.call("hriemo_rowsum_f32", _p(part), _p(dbeta_h), B, nc * d, st)               # d loss / d beta through the fusion
        dbeta2 = dbeta.contiguous().float() if dbeta is not None else None
        dpre16 = torch.empty((B, 1), dtype=BF16, device=dev)
        # gate_dpre with d = 1: (dbeta_h + dbeta) * beta * (1 - beta)
        _lib.call("hriemo_gate_dpre", _p(dbeta_h), 1, _p(dbeta2), _p(beta), _p(dpre16), B, 1, st)
        dpre = dpre16.float().view(B)                                                      # [B]-sized plumbing
        Hd = hid.shape[1]
        dhid = torch.empty((B, Hd), dtype=BF16, device=dev)
        dw2 = torch.empty((1, Hd), **f32)
        db2 = torch.empty(1, **f32)
        _lib.call("hriemo_rowdot_bwd", _p(dpre), _p(hid), None, _p(w2f), _p(dhid), _p(dw2), _p(db2), 0, B, Hd, st)
        dhid = dhid * (hid > 0)                                                            # ReLU mask, [B,h]-sized plumbing
        dw1 = torch.empty((Hd, 4 * d), **f32)
        linear_dw(dhid, gin, dw1)
        db1 = torch.empty(Hd, **f32)
        colsum(dhid, db1)
        dgin = linear_dx(dhid, w1_16)
        dpa, dpt = torch.empty((B, d), **f32), torch.empty((B, d), **f32)
        ones = torch.ones((B, 2), **f32)           # scalar_gate_dx divides by the valid counts itself
        _lib.call("hriemo_gate_input_bwd", _p(dgin), _p(a_pool), _p(t_pool), _p(ones), _p(dpa), _p(dpt), B, d, st)
        bflat = beta.reshape(B).contiguous()
        dxa = torch.empty((B, La, d), dtype=BF16, device=dev)
        dxt = torch.empty((B, Lt, d), dtype=BF16, device=dev)
        _lib.call("hriemo_scalar_gate_dx", _p(dH2), L, _p(bflat), 1, _p(dpa), _p(cnt_a), _p(kpm_a), _p(dxa), B, La, d, st)
        _lib.call("hriemo_scalar_gate_dx", _p(dH2), L, _p(bflat), 0, _p(dpt), _p(cnt_t), _p(kpm_t), _p(dxt), B, Lt, d, st)
        return dxa, dxt, dw1, db1, dw2, db2, None, None, None


class FusionLossFn(torch.autograd.Function):
    """Trainer loss with its gradients from ONE kernel (hriemo_fusion_loss): mean BCEWithLogits(logits, targets; pos_weight) +
    reg(beta).  reg_mode 1 = -coef*mean(beta(1-beta)) (train_fusion_seq_level_decoder.py:318-326), 2 = +coef*entropy(beta)
    (train_mosei_fusion_seq_level_decoder.py:340-347,385-386); `scale` = 1/grad_accum (:387)."""

    @staticmethod
    def forward(ctx, logits, beta, targets, pos_weight, reg_mode, reg_coef, scale):
        _require_gpu(logits)
        B, Ne = logits.shape
        x, y = logits.contiguous().float(), targets.contiguous().float()
        bt = beta.contiguous().float().view(B) if beta is not None else None
        pw = pos_weight.contiguous().float() if pos_weight is not None else None
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        dx = torch.empty((B, Ne), dtype=torch.float32, device=x.device)
        db = torch.empty(B, dtype=torch.float32, device=x.device) if bt is not None else None
        _lib.call("hriemo_fusion_loss", _p(x), _p(y), _p(pw), _p(bt), B, Ne, int(reg_mode) if bt is not None else 0, float(reg_coef),
                  float(scale), _p(loss), _p(dx), _p(db), _stream())
        ctx.save_for_backward(dx, db)
        ctx.shapes = (logits.shape, beta.shape if beta is not None else None, logits.dtype, beta.dtype if beta is not None else None)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        dx, db = ctx.saved_tensors
        ls, bs, ld, bd = ctx.shapes
        if _is_one(g):                       # loss.backward(gradient=_ops.one(device)): the kernel's gradients are the answer
            return dx.to(ld).view(ls), (db.to(bd).view(bs) if db is not None else None), None, None, None, None, None
        gl = (dx * g).to(ld).view(ls)
        gb = (db * g).to(bd).view(bs) if db is not None else None
        return gl, gb, None, None, None, None, None


class FusionLossCEFn(torch.autograd.Function):
    """Single-label trainer loss (train_fusion_seq_level_decoder.py:312-314,325-326,413-414): CrossEntropyLoss(logits, labels) +
    reg(beta), value and gradients from ONE kernel (hriemo_fusion_loss_ce)."""

    @staticmethod
    def forward(ctx, logits, beta, labels, reg_mode, reg_coef, scale):
        _require_gpu(logits)
        B, C = logits.shape
        x = logits.contiguous().float()
        lb = labels.contiguous().to(torch.int64)
        bt = beta.contiguous().float().view(B) if beta is not None else None
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        dx = torch.empty((B, C), dtype=torch.float32, device=x.device)
        db = torch.empty(B, dtype=torch.float32, device=x.device) if bt is not None else None
        _lib.call("hriemo_fusion_loss_ce", _p(x), _p(lb), _p(bt), B, C, int(reg_mode) if bt is not None else 0, float(reg_coef),
                  float(scale), _p(loss), _p(dx), _p(db), _stream())
        ctx.save_for_backward(dx, db)
        ctx.shapes = (logits.shape, beta.shape if beta is not None else None, logits.dtype, beta.dtype if beta is not None else None)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        dx, db = ctx.saved_tensors
        ls, bs, ld, bd = ctx.shapes
        if _is_one(g):
            return dx.to(ld).view(ls), (db.to(bd).view(bs) if db is not None else None), None, None, None, None
        gl = (dx * g).to(ld).view(ls)
        gb = (db * g).to(bd).view(bs) if db is not None else None
        return gl, gb, None, None, None, None


class LinearFn(_GradModeAware, torch.autograd.Function):
    """y[..., N] (fp32) = x[..., K] . W[N,K]^T + b for any K (MOSEI projections: K = 74 / 300, padded to a
    multiple of 8 internally) -- models/mosei_fusion_with_emotion_decoder.py:41-42,63-65."""

    @staticmethod
    def forward(ctx, x, w, b, sh):
        if precision() == "fp32":
            return _fp32().linear_any_k(ctx, x, w, b, sh)
        _require_fp32_masters(w, b)
        _require_gpu(x)
        K = x.shape[-1]
        N = w.shape[0]
        M = x.numel() // K
        kp = (K + 7) // 8 * 8
        xb = torch.zeros((M, kp), dtype=BF16, device=x.device)
        xb[:, :K].copy_(x.reshape(M, K))
        w16 = padded_shadow(sh, w, kp)
        y = linear_fwd(xb, w16, b, out_f32=True)
        ctx.save_for_backward(xb, w16)
        ctx.cfg = (tuple(x.shape), x.dtype, K, N, M, kp)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        if getattr(ctx, "fp32", False):
            return _fp32().linear_any_k_bwd(ctx, dy)
        xb, w16 = ctx.saved_tensors
        shape, xdtype, K, N, M, kp = ctx.cfg
        dyb = dy.reshape(M, N).to(BF16).contiguous()
        dx = linear_dx(dyb, w16)[:, :K].to(xdtype).reshape(shape) if ctx.needs_input_grad[0] else None
        dwp = torch.empty((N, kp), dtype=torch.float32, device=dyb.device)
        linear_dw(dyb, xb, dwp)
        db = torch.empty(N, dtype=torch.float32, device=dyb.device)
        colsum(dyb, db)
        return dx, dwp[:, :K].contiguous(), db, None


class ExpandFn(_GradModeAware, torch.autograd.Function):
    """queries[N_e,d] -> [B,N_e,d] bf16   (models/emotion_decoder.py:127)"""

    @staticmethod
    def forward(ctx, q, B, twin=False):
        """twin: also return the fp32 copy [B,N_e,d] (the residual twin of the first decoder layer; not differentiable)"""
        _require_gpu(q)
        Ne, d = q.shape
        out = torch.empty((B, Ne, d), dtype=BF16, device=q.device)
        out32 = torch.empty((B, Ne, d), dtype=torch.float32, device=q.device) if twin else None
        src = q.detach().float().contiguous()
        _lib.call("hriemo_expand_rows", _p(src), _p(out), _p(out32), B, Ne * d, _stream())
        ctx.cfg = (B, Ne, d)
        ctx.params = (q,)
        ctx.fp32 = precision() == "fp32"      # fp32 mode: the decoder reads the fp32 copy and its gradient comes back through it
        if ctx.fp32:
            ctx.set_materialize_grads(False)
        if twin:
            if not ctx.fp32:
                ctx.mark_non_differentiable(out32)
            return out, out32
        return out

    @staticmethod
    def backward(ctx, dout, _d32=None):
        B, Ne, d = ctx.cfg
        if ctx.fp32:
            return _fp32().expand_bwd(dout, _d32, B, Ne, d, ctx.params[0]), None, None
        g = _contig_bf16(dout).view(B, Ne * d)
        sink = GradSink(ctx.params)
        dq = sink.buf(ctx.params[0])
        colsum(g, dq.view(Ne * d), sink.fused)
        sink.done(after_flush=ctx.params)
        return sink.ret(dq), None, None


class RowDotFn(_GradModeAware, torch.autograd.Function):
    """logits[M] = z[M,d] . w[1,d] + b   (models/emotion_decoder.py:155)"""

    @staticmethod
    def forward(ctx, z, z32, w, b):
        _require_fp32_masters(w, b)
        _require_gpu(z)
        B, Ne, d = z.shape
        z2 = _contig_bf16(z).view(B * Ne, d)
        z32 = _c32(z32)
        z32v = z32.view(B * Ne, d) if z32 is not None else None
        wf = w.detach().float().contiguous()
        bf = b.detach().float().contiguous()
        out = torch.empty(B * Ne, dtype=torch.float32, device=z.device)
        _lib.call("hriemo_rowdot_fwd", _p(z2), _p(z32v), _p(wf), _p(bf), _p(out), B * Ne, d, _stream())
        ctx.save_for_backward(z2, z32v, wf)
        ctx.cfg = (B, Ne, d)
        ctx.params = (w, b)
        ctx.fp32 = precision() == "fp32" and z32v is not None
        return out.view(B, Ne)

    @staticmethod
    def backward(ctx, dl):
        z2, z32v, wf = ctx.saved_tensors
        B, Ne, d = ctx.cfg
        if ctx.fp32:                            # gradient of z in fp32, through the fp32 slot
            dz32, dw, db = _fp32().rowdot_bwd(dl, z32v, wf, B, Ne, d)
            return None, dz32, dw.view_as(ctx.params[0]), db.view_as(ctx.params[1])
        dl2 = dl.contiguous().float().view(-1)
        dz = torch.empty((B * Ne, d), dtype=BF16, device=z2.device)
        sink = GradSink(ctx.params)
        dw, db = sink.buf(ctx.params[0]), sink.buf(ctx.params[1])
        _lib.call("hriemo_rowdot_bwd", _p(dl2), _p(z2), _p(z32v), _p(wf), _p(dz), _p(dw), _p(db), int(sink.fused), B * Ne, d, _stream())
        sink.done()
        return dz.view(B, Ne, d), None, sink.ret(dw), sink.ret(db)
