"""fp32-tolerance path (HRIEMO_PRECISION=fp32 / hri_emo_amd.set_precision("fp32")): forward AND backward.

The reference computes in fp32 throughout, and its IEMOCAP trainer runs it that way (no autocast:
scripts/fusion/train_fusion_seq_level_decoder.py:310-334); the bf16 product path meets north_star's 1e-2, this mode 1e-3 on the same
golden fixtures -- outputs (tests/test_gpu_fp32_mode.py) and, since round 4, every parameter's gradient of the training step.
The kernels are csrc/fp32mode.hip: exact three-way bf16 operand split for every Linear (forward, dX and dW: six bf16 products per
fp32 product on the bf16 GEMM kernel over a 6x longer contraction, fp32 accumulate and output), fp32 MFMA attention cores (forward, dQ, dK / dV),
fp32 row kernels for LayerNorm, the gate and the head.  This file is the host-side composition: one forward and one backward
function per sub-layer Function of _ops.py, which dispatches here and stays the autograd node.

Activations travel as fp32 tensors: every sub-layer still returns the (bf16 copy, fp32) pair of the product path, consumers read
the fp32 member, and gradients flow back through the fp32 slots.  Dropout (the reference's trainer keeps the modules' default 0.1):
the fp32 kernels drop where the bf16 ones do -- attention weights, sub-layer outputs before the residual add, the decoder FFN's
hidden layer -- from the same counter hash with the same (seed, site, offset) keys, so a model draws the same masks in either
precision; the backward replays them, nothing is stored.

Reference arithmetic: models/cross_modal_block_tacfn.py:70-125, models/beta_gate_tacfn.py:68-118, models/emotion_decoder.py:30-64,
116-162 (all fp32 nn.Modules)."""
import torch

from . import _lib, _ops

F32 = torch.float32
BF16 = torch.bfloat16


def recording(ctx):
    """autograd will call this node's backward (_ops._GradModeAware notes the grad mode at apply() time)"""
    return _ops.recording(ctx)


def _drop_args(drop, dev):
    """(p, seed, site, offset) | None -> the five dropout arguments of an fp32 kernel"""
    if drop is None or drop[0] <= 0:
        return 0.0, 0, None, 0, 0
    p, seed, site, off = drop
    return float(p), seed, _ops._p(_ops.seed_word(dev)), site, off


def _on(drop):
    return drop is not None and drop[0] > 0


def tag32(t16, t32):
    """the fp32 twin travels with a bf16 tensor that crosses a module boundary without a pair slot (inference only)"""
    t16._hriemo_f32 = t32
    return t16


def f32_of(t):
    t32 = getattr(t, "_hriemo_f32", None)
    if t32 is not None and t32.shape == t.shape:
        return t32
    return t if t.dtype == F32 else t.float()


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _new(shape, like, dtype=F32):
    return torch.empty(shape, dtype=dtype, device=like.device)


def _ws(nbytes, dev):
    return _ops.workspace(nbytes, dev, slot=2)


# ----------------------------------------------------------------------------- operand splits and the three GEMMs of a Linear
def split(x32, form, relu=False, mask=None):
    """x32 [M,K] fp32 (row stride free) -> bf16 split operand, form 0 / 1: [M,3K] = [hi|mid|hi] / [hi|hi|mid] (contraction along
    the columns), form 2 / 3: [3M,K] = [hi;mid;hi] / [hi;hi;mid] (contraction along the rows); forms 4 / 5 and 6 / 7: the same with
    the exact three-way split x = hi + mid + lo and six blocks ([hi|mid|lo|hi|mid|hi] / [hi|hi|hi|mid|mid|lo]: what this module
    uses); relu: max(x, 0) first; mask (fp32 [M,K]): x * (mask > 0) first"""
    M, K = x32.shape
    if K % 8:
        raise ValueError(f"fp32 mode: width {K} must be a multiple of 8")
    shape = {0: (M, 3 * K), 1: (M, 3 * K), 2: (3 * M, K), 3: (3 * M, K), 4: (M, 6 * K), 5: (M, 6 * K), 6: (6 * M, K), 7: (6 * M, K)}[form]
    y = _new(shape, x32, BF16)
    _lib.call("hriemo_split3_f32", _ops._p(x32), x32.stride(0), M, K, _ops._p(y), form, int(relu), _ops._p(mask),
              mask.stride(0) if mask is not None else 0, _ops._stream())
    return y


def split3(x32, relu=False, layout=0):
    """(forward name) [M,3K]: layout 0 activations [hi|mid|hi], 1 weights [hi|hi|mid]"""
    return split(x32, layout, relu=relu)


def _weight_split(sh, w, form, rows=None, kp=None):
    """split copy of an fp32 master weight (form 5: [N,6K] for the forward, form 7: [6N,K] for dX), cached next to the bf16 shadows
    and refreshed like them; rows = (r0, r1): of that row block only (the Q or K|V rows of a packed in-projection)"""
    key = ("x3", id(w), form, rows, kp)
    ent = sh._d.get(key)
    ver = (w._version, w.data_ptr(), _ops.WEIGHTS_EPOCH)
    if ent is None or ent[0] != ver or ent[1].device != w.device:
        _ops._require_gpu(w)
        _ops._require_fp32_master(w)
        src = _c(w.detach())
        if rows is not None:
            src = src[rows[0]:rows[1]]
        if kp is not None and kp != src.shape[1]:
            pad = torch.zeros((src.shape[0], kp), dtype=F32, device=src.device)
            pad[:, :src.shape[1]].copy_(src)
            src = pad
        ent = (ver, split(src, form))
        sh._d[key] = ent
    return ent[1]


def linear(x32, sh, w, bias, rows=None, relu_in=False, kp=None):
    """y32[M,N] = (relu_in ? relu(x32) : x32) . W[rows]^T + b[rows] to fp32 accuracy: both operands split into three bf16 parts
    that add up to the fp32 value exactly, six bf16 products per fp32 product (everything but mid.lo, lo.mid, lo.lo: 2^-24).
    The forward needs that: a pre-activation that is 4e-6 off (the three-product form the backward GEMMs use) can sit on the
    other side of a ReLU than the reference's, and a flipped unit is a gradient error of its whole weight row."""
    M, K = x32.shape
    ws = _weight_split(sh, w, 5, None, kp)             # [N, 6K]
    b = bias.detach() if bias is not None else None
    if rows is not None:
        ws = ws[rows[0]:rows[1]]
        b = b[rows[0]:rows[1]] if b is not None else None
    N = ws.shape[0]
    xs = split(x32, 4, relu=relu_in)                   # [M, 6K]
    y = _new((M, N), x32)
    _ops.gemm(0, 0, M, N, 6 * K, xs, 6 * K, ws, ws.stride(0), y, N, c_f32=True, bias=b)
    return y


def linear_dx(dy32, sh, w, rows=None, mask=None, into=None, kp=None):
    """dX[M,K] = (dY * (mask > 0)) . W[rows]; into: an fp32 [M,K] tensor the product is ADDED to (the residual path's gradient)"""
    M, N = dy32.shape
    wr = _weight_split(sh, w, 7, rows, kp)             # [6N, K]
    K = wr.shape[1]
    a = split(dy32, 4, mask=mask)                      # [M, 6N]
    out = into if into is not None else _new((M, K), dy32)
    _ops.gemm(0, 1, M, K, 6 * N, a, 6 * N, wr, K, out, K, c_f32=True, accumulate=into is not None)
    return out


def linear_dw(dy32, x32, mask=None, relu_x=False):
    """dW[N,K] = (dY * (mask > 0))^T . (relu_x ? relu(X) : X): the contraction runs over the rows, both operands split along them"""
    M, N = dy32.shape
    K = x32.shape[1]
    a = split(dy32, 6, mask=mask)                      # [6M, N]
    b = split(x32, 7, relu=relu_x)                     # [6M, K]
    out = _new((N, K), dy32)
    _ops.gemm(1, 1, N, K, 6 * M, a, N, b, K, out, K, c_f32=True)
    return out


def colsum(x32, mask=None):
    """[N] = column sums of x32 [M,N] (times (mask > 0)): bias gradients"""
    M, N = x32.shape
    out = _new((N,), x32)
    ws = _ws(_lib.lib().hriemo_colsum_f32_workspace_bytes(M, N), x32.device)
    _lib.call("hriemo_colsum_f32", _ops._p(x32), x32.stride(0), M, N, _ops._p(mask), mask.stride(0) if mask is not None else 0,
              _ops._p(out), 0, _ops._p(ws), _ops._stream())
    return out


# ----------------------------------------------------------------------------- attention cores, LayerNorm
def attn(q, k, v, B, H, Lq, Lk, hd, kpm, want_lse=False, drop=None):
    """drop = (p, seed, site, b_offset): dropout on the attention weights (the keys of _ops.attn_fwd)"""
    o = _new((B * Lq, H * hd), q)
    lse = _new((B, H, Lq), q) if want_lse else None
    if _on(drop) and _ops.DROP_LOG is not None:
        _ops.DROP_LOG.append(("attn", drop[1], drop[2], B, H, Lq, Lk, float(drop[0]), drop[3]))
    _lib.call("hriemo_attn_fwd_f32", _ops._p(q), q.stride(0), _ops._p(k), k.stride(0), _ops._p(v), v.stride(0), _ops._p(o), H * hd,
              _ops._p(kpm), _ops._p(lse), B, H, Lq, Lk, hd, *_drop_args(drop, q.device), _ops._stream())
    return o, lse


def attn_bwd(q, k, v, o, do, lse, dq, dk, dv, B, H, Lq, Lk, hd, kpm, drop=None):
    delta = _new((B, H, Lq), q)
    _lib.call("hriemo_attn_bwd_f32", _ops._p(q), q.stride(0), _ops._p(k), k.stride(0), _ops._p(v), v.stride(0), _ops._p(o), o.stride(0),
              _ops._p(do), do.stride(0), _ops._p(kpm), _ops._p(lse), _ops._p(dq), dq.stride(0), _ops._p(dk), dk.stride(0), _ops._p(dv),
              dv.stride(0), _ops._p(delta), B, H, Lq, Lk, hd, *_drop_args(drop, q.device), _ops._stream())


def probs(q, k, B, H, Lq, Lk, hd, kpm, lse, drop=None):
    p = _new((B, Lq, Lk), q)
    _lib.call("hriemo_attn_probs_f32", _ops._p(q), q.stride(0), _ops._p(k), k.stride(0), _ops._p(kpm), _ops._p(lse), _ops._p(p), B, H,
              Lq, Lk, hd, *_drop_args(drop, q.device), _ops._stream())
    return p


def add_ln(g32, x32, gamma, beta, want16=True, drop=None):
    """LayerNorm(x32 + drop(g32)) (x32 may be None) -> (bf16 copy | None, fp32); drop = (p, seed, site, row_offset) as
    _ops.add_ln_fwd"""
    M, d = g32.shape
    y32 = _new((M, d), g32)
    y16 = _new((M, d), g32, BF16) if want16 else None
    if _on(drop) and _ops.DROP_LOG is not None:
        _ops.DROP_LOG.append(("rows", drop[1], drop[2], M, d, float(drop[0]), drop[3]))
    _lib.call("hriemo_add_ln_f32", _ops._p(g32), _ops._p(x32), _ops._p(gamma.detach()), _ops._p(beta.detach()), _ops._p(y32),
              _ops._p(y16), M, d, _ops._EPS, *_drop_args(drop, g32.device), _ops._stream())
    return y16, y32


def add_ln_bwd(dy32, g32, x32, gamma, want_dbias=True, drop=None):
    """backward of LayerNorm(x32 + drop(g32)): -> (dS [M,d] = gradient of the sum = dX, dG = gradient of g32 (dS itself without
    dropout), dgamma, dbeta, dbias = colsum(dG) | None)"""
    M, d = g32.shape
    ds = _new((M, d), g32)
    dg = _new((M, d), g32) if _on(drop) else None
    stats = _new((3, d), g32)
    ws = _ws(_lib.lib().hriemo_add_ln_bwd_f32_workspace_bytes(M, d), g32.device)
    _lib.call("hriemo_add_ln_bwd_f32", _ops._p(dy32), _ops._p(g32), _ops._p(x32), _ops._p(gamma.detach()), _ops._p(ds), _ops._p(dg),
              _ops._p(stats[0]), _ops._p(stats[1]), _ops._p(stats[2]) if want_dbias else None, 0, M, d, _ops._EPS,
              *_drop_args(drop, g32.device), _ops._p(ws), _ops._stream())
    return ds, (dg if dg is not None else ds), stats[0], stats[1], (stats[2] if want_dbias else None)


def dropout(x32, drop, relu=False, gate=None, log=True):
    """drop(relu ? max(x, 0) : x) [* (gate > 0)]: the FFN's hidden dropout (forward: relu; backward: x = gradient, gate = the
    pre-activations, log=False)"""
    M, N = x32.shape
    y = _new((M, N), x32)
    if log and _on(drop) and _ops.DROP_LOG is not None:
        _ops.DROP_LOG.append(("rows", drop[1], drop[2], M, N, float(drop[0]), drop[3]))
    _lib.call("hriemo_dropout_f32", _ops._p(_c(x32)), _ops._p(y), M, N, int(relu), _ops._p(gate), *_drop_args(drop, x32.device),
              _ops._stream())
    return y


def _twin(x, x32):
    return _c(x32) if x32 is not None else f32_of(_c(x))


def _total(dy16, dy32, shape):
    """incoming gradient of a (bf16 copy, fp32) output pair, as one contiguous fp32 [M,d] tensor"""
    if dy32 is None and dy16 is None:
        return None
    if dy32 is None:
        g = dy16.float()
    elif dy16 is None:
        g = dy32 if dy32.dtype == F32 else dy32.float()
    else:
        g = dy32.float() + dy16.float()
    return _c(g).view(shape)


def _route(ctx, dx32, x_slot, x32_slot, shape):
    """hand the input gradient to the slot the fp32 values came from: the twin when one was given, else the bf16 / caller tensor"""
    out = [None, None]
    if ctx.f32_from_twin:
        out[1] = dx32.view(shape) if ctx.needs_input_grad[x32_slot] else None
    else:
        out[0] = dx32.view(shape).to(ctx.f32_x_dtype) if ctx.needs_input_grad[x_slot] else None
    return out


# ----------------------------------------------------------------------------- sub-layers (same results tuple as the Functions)
def _drops(p, seed, site, b_off, rows_per_sample):
    """the two dropout sites of an attention sub-layer, keyed as the bf16 Functions key them (_ops.SelfAttnLN.forward)"""
    if p <= 0:
        return None, None
    return (p, seed, site, b_off), (p, seed, site + 1, b_off * rows_per_sample)


def self_attn_ln(ctx, x, x32, w_in, b_in, w_out, b_out, gamma, beta, sh, H, kpm, need_w, p=0.0, seed=0, site=0, b_off=0):
    _ops._require_fp32_masters(w_in, b_in, w_out, b_out, gamma, beta)
    _ops._require_gpu(x)
    B, L, d = x.shape
    hd = _ops._heads(d, H)
    rec = recording(ctx)
    d_attn, d_res = _drops(p, seed, site, b_off, L)
    xf = _twin(x, x32).view(B * L, d)
    qkv = linear(xf, sh, w_in, b_in)
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    o, lse = attn(q, k, v, B, H, L, L, hd, kpm, want_lse=need_w or rec, drop=d_attn)
    g = linear(o, sh, w_out, b_out)
    y16, y32 = add_ln(g, xf, gamma, beta, drop=d_res)
    pr = probs(q, k, B, H, L, L, hd, kpm, lse, drop=d_attn) if need_w else None
    ctx.fp32 = True
    if rec:
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(xf, qkv, o, lse, g, kpm)
        ctx.f32_cfg = (B, L, d, H, hd)
        ctx.f32_drop = (d_attn, d_res)
        ctx.f32_params = (w_in, b_in, w_out, b_out, gamma, beta, sh)
        ctx.f32_from_twin, ctx.f32_x_dtype = x32 is not None, x.dtype
        if pr is not None:
            ctx.mark_non_differentiable(pr)
    return y16.view(B, L, d), y32.view(B, L, d), pr


def self_attn_ln_bwd(ctx, dy, dy32):
    xf, qkv, o, lse, g, kpm = ctx.saved_tensors
    B, L, d, H, hd = ctx.f32_cfg
    w_in, b_in, w_out, b_out, gamma, beta, sh = ctx.f32_params
    M = B * L
    d_attn, d_res = ctx.f32_drop
    dyt = _total(dy, dy32, (M, d))
    ds, dg, dgamma, dbeta, db_out = add_ln_bwd(dyt, g, xf, gamma, drop=d_res)
    dw_out = linear_dw(dg, o)
    do = linear_dx(dg, sh, w_out)
    dqkv = _new((M, 3 * d), xf)
    attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, do, lse, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, H, L, L, hd, kpm,
             drop=d_attn)
    dw_in = linear_dw(dqkv, xf)
    db_in = colsum(dqkv)
    dx = linear_dx(dqkv, sh, w_in, into=ds)            # + the residual path's gradient
    gx = _route(ctx, dx, 0, 1, (B, L, d))
    return (gx[0], gx[1], dw_in, db_in, dw_out, db_out, dgamma, dbeta) + (None,) * 8


def cross_attn_ln(ctx, xq, xq32, xkv, w_in, b_in, w_out, b_out, gamma, beta, sh, H, kpm, need_w, p=0.0, seed=0, site=0, b_off=0):
    _ops._require_fp32_masters(w_in, b_in, w_out, b_out, gamma, beta)
    _ops._require_gpu(xq)
    B, Lq, d = xq.shape
    Lk = xkv.shape[1]
    hd = _ops._heads(d, H)
    rec = recording(ctx)
    d_attn, d_res = _drops(p, seed, site, b_off, Lq)
    xqf = _twin(xq, xq32).view(B * Lq, d)
    xkvf = _c(f32_of(xkv)).view(B * Lk, d)
    q = linear(xqf, sh, w_in, b_in, rows=(0, d))
    kv = linear(xkvf, sh, w_in, b_in, rows=(d, 3 * d))
    k, v = kv[:, :d], kv[:, d:]
    o, lse = attn(q, k, v, B, H, Lq, Lk, hd, kpm, want_lse=need_w or rec, drop=d_attn)
    g = linear(o, sh, w_out, b_out)
    y16, y32 = add_ln(g, xqf, gamma, beta, drop=d_res)
    pr = probs(q, k, B, H, Lq, Lk, hd, kpm, lse, drop=d_attn) if need_w else None
    ctx.fp32 = True
    if rec:
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(xqf, xkvf, q, kv, o, lse, g, kpm)
        ctx.f32_cfg = (B, Lq, Lk, d, H, hd)
        ctx.f32_drop = (d_attn, d_res)
        ctx.f32_params = (w_in, b_in, w_out, b_out, gamma, beta, sh)
        ctx.f32_from_twin, ctx.f32_x_dtype, ctx.f32_kv_dtype = xq32 is not None, xq.dtype, xkv.dtype
        if pr is not None:
            ctx.mark_non_differentiable(pr)
    return y16.view(B, Lq, d), y32.view(B, Lq, d), pr


def cross_attn_ln_bwd(ctx, dy, dy32):
    xqf, xkvf, q, kv, o, lse, g, kpm = ctx.saved_tensors
    B, Lq, Lk, d, H, hd = ctx.f32_cfg
    w_in, b_in, w_out, b_out, gamma, beta, sh = ctx.f32_params
    d_attn, d_res = ctx.f32_drop
    dyt = _total(dy, dy32, (B * Lq, d))
    ds, dg, dgamma, dbeta, db_out = add_ln_bwd(dyt, g, xqf, gamma, drop=d_res)
    dw_out = linear_dw(dg, o)
    do = linear_dx(dg, sh, w_out)
    dq = _new((B * Lq, d), xqf)
    dkv = _new((B * Lk, 2 * d), xqf)
    attn_bwd(q, kv[:, :d], kv[:, d:], o, do, lse, dq, dkv[:, :d], dkv[:, d:], B, H, Lq, Lk, hd, kpm, drop=d_attn)
    dw_in = _new((3 * d, d), xqf)
    dw_in[:d].copy_(linear_dw(dq, xqf))
    dw_in[d:].copy_(linear_dw(dkv, xkvf))
    db_in = torch.cat([colsum(dq), colsum(dkv)])
    dxq = linear_dx(dq, sh, w_in, rows=(0, d), into=ds)
    gx = _route(ctx, dxq, 0, 1, (B, Lq, d))
    dxkv = None
    if ctx.needs_input_grad[2]:
        dxkv = linear_dx(dkv, sh, w_in, rows=(d, 3 * d)).view(B, Lk, d).to(ctx.f32_kv_dtype)
    return (gx[0], gx[1], dxkv, dw_in, db_in, dw_out, db_out, dgamma, dbeta) + (None,) * 12


def ffn_ln(ctx, x, x32, w1, b1, w2, b2, gamma, beta, sh, p=0.0, p_mid=0.0, seed=0, site=0, b_off=0):
    _ops._require_fp32_masters(w1, b1, w2, b2, gamma, beta)
    _ops._require_gpu(x)
    shape = x.shape
    d = shape[-1]
    L = shape[1] if len(shape) == 3 else 1
    d_res = (p, seed, site + 1, b_off * L) if p > 0 else None
    d_mid = (p_mid, seed, site + 2, b_off * L) if p_mid > 0 else None        # keys of _ops.FFNLN.forward
    xf = _twin(x, x32).view(-1, d)
    h = linear(xf, sh, w1, b1)
    if d_mid is not None:
        g = linear(dropout(h, d_mid, relu=True), sh, w2, b2)
    else:
        g = linear(h, sh, w2, b2, relu_in=True)      # ReLU applied while the hidden activations are split
    y16, y32 = add_ln(g, xf, gamma, beta, drop=d_res)
    ctx.fp32 = True
    if recording(ctx):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(xf, h, g)
        ctx.f32_cfg = tuple(shape)
        ctx.f32_drop = (d_res, d_mid)
        ctx.f32_params = (w1, b1, w2, b2, gamma, beta, sh)
        ctx.f32_from_twin, ctx.f32_x_dtype = x32 is not None, x.dtype
    return y16.view(shape), y32.view(shape)


def ffn_ln_bwd(ctx, dy, dy32):
    xf, h, g = ctx.saved_tensors
    shape = ctx.f32_cfg
    w1, b1, w2, b2, gamma, beta, sh = ctx.f32_params
    M, d = xf.shape
    d_res, d_mid = ctx.f32_drop
    dyt = _total(dy, dy32, (M, d))
    ds, dg, dgamma, dbeta, db2 = add_ln_bwd(dyt, g, xf, gamma, drop=d_res)
    if d_mid is not None:                             # g = drop(relu(h)) . W2^T + b2: the dropped activations are rebuilt, not stored
        dw2 = linear_dw(dg, dropout(h, d_mid, relu=True, log=False))
        da = dropout(linear_dx(dg, sh, w2), d_mid, gate=h, log=False)     # * keep / (1 - p) * relu'(h)
        dw1 = linear_dw(da, xf)
        db1 = colsum(da)
        dx = linear_dx(da, sh, w1, into=ds)
    else:
        dw2 = linear_dw(dg, h, relu_x=True)           # g = relu(h) . W2^T + b2
        da = linear_dx(dg, sh, w2)                    # gradient of relu(h); ReLU's derivative is applied where da is consumed
        dw1 = linear_dw(da, xf, mask=h)
        db1 = colsum(da, mask=h)
        dx = linear_dx(da, sh, w1, mask=h, into=ds)
    gx = _route(ctx, dx, 0, 1, shape)
    return (gx[0], gx[1], dw1, db1, dw2, db2, dgamma, dbeta) + (None,) * 7


def beta_gate(ctx, h_a, h_a32, h_t, h_t32, ga, ba, gt, bt, w1, b1, w2, b2, sh, kpm_a, kpm_t):
    """models/beta_gate_tacfn.py:68-118 in fp32 -> (h_fusion fp32 [B,L,d], beta [B,1]).  h_fusion is handed on as the fp32 tensor
    itself (the decoder reads it as its memory and autograd sees it); ctx None: plain forward, nothing saved"""
    _ops._require_fp32_masters(ga, ba, gt, bt, w1, b1, w2, b2)
    _ops._require_gpu(h_a)
    B, La, d = h_a.shape
    Lt = h_t.shape[1]
    L = La if La == Lt else Lt
    if La < L:
        raise RuntimeError(f"BetaGate: audio length {La} < text length {Lt}; the reference cannot fuse this either")
    dev = h_a.device
    st = _ops._stream()
    a32 = _twin(h_a, h_a32).view(B * La, d)
    t32 = _twin(h_t, h_t32).view(B * Lt, d)
    _, An = add_ln(a32, None, ga, ba, want16=False)
    _, Tn = add_ln(t32, None, gt, bt, want16=False)
    a_pool = torch.empty((B, d), dtype=F32, device=dev)
    t_pool = torch.empty((B, d), dtype=F32, device=dev)
    _lib.call("hriemo_masked_mean_f32", _ops._p(An), _ops._p(kpm_a), _ops._p(a_pool), B, La, d, st)
    _lib.call("hriemo_masked_mean_f32", _ops._p(Tn), _ops._p(kpm_t), _ops._p(t_pool), B, Lt, d, st)
    gin = torch.empty((B, 4 * d), dtype=F32, device=dev)
    _lib.call("hriemo_gate_input_f32", _ops._p(a_pool), _ops._p(t_pool), _ops._p(gin), B, d, st)
    hid = linear(gin, sh, w1, b1)
    pre = linear(hid, sh, w2, b2, relu_in=True)
    w = torch.empty((B, d), dtype=F32, device=dev)
    beta = torch.empty((B, 1), dtype=F32, device=dev)
    _lib.call("hriemo_sigmoid_beta_f32", _ops._p(pre), _ops._p(w), _ops._p(beta), B, d, st)
    H32 = torch.empty((B, L, d), dtype=F32, device=dev)
    _lib.call("hriemo_fuse_f32", _ops._p(w), _ops._p(An), La, _ops._p(Tn), Lt, _ops._p(H32), None, B, L, d, st)
    if ctx is not None:
        ctx.fp32 = True
        if recording(ctx):
            ctx.set_materialize_grads(False)
            ctx.save_for_backward(a32, t32, An, Tn, a_pool, t_pool, gin, hid, w, kpm_a, kpm_t)
            ctx.f32_cfg = (B, La, Lt, L, d)
            ctx.f32_params = (ga, ba, gt, bt, w1, b1, w2, b2, sh)
            ctx.f32_twins = (h_a32 is not None, h_t32 is not None, h_a.dtype, h_t.dtype)
    return H32, beta


def beta_gate_bwd(ctx, dH, dbeta):
    a32, t32, An, Tn, a_pool, t_pool, gin, hid, w, kpm_a, kpm_t = ctx.saved_tensors
    B, La, Lt, L, d = ctx.f32_cfg
    ga, ba, gt, bt, w1, b1, w2, b2, sh = ctx.f32_params
    dev = a32.device
    st = _ops._stream()
    dH = _c(dH.float()) if dH is not None else torch.zeros((B, L, d), dtype=F32, device=dev)
    db = _c(dbeta.float()).view(B) if dbeta is not None else None
    dpre = torch.empty((B, d), dtype=F32, device=dev)
    _lib.call("hriemo_gate_dpre_f32", _ops._p(dH), _ops._p(An), La, _ops._p(Tn), Lt, _ops._p(w), _ops._p(db), _ops._p(dpre), B, L, d, st)
    # MLP: pre = relu(hid) . W2^T + b2, hid = gin . W1^T + b1
    dw2 = linear_dw(dpre, hid, relu_x=True)
    db2 = colsum(dpre)
    dhid = linear_dx(dpre, sh, w2)
    dw1 = linear_dw(dhid, gin, mask=hid)
    db1 = colsum(dhid, mask=hid)
    dgin = linear_dx(dhid, sh, w1, mask=hid)
    da = torch.empty((B, d), dtype=F32, device=dev)
    dt = torch.empty((B, d), dtype=F32, device=dev)
    _lib.call("hriemo_gate_input_bwd_f32", _ops._p(dgin), _ops._p(a_pool), _ops._p(t_pool), _ops._p(da), _ops._p(dt), B, d, st)
    outs = []
    for is_a, dpool, kpm, x32, gamma, Lx in ((1, da, kpm_a, a32, ga, La), (0, dt, kpm_t, t32, gt, Lt)):
        dY = torch.empty((B * Lx, d), dtype=F32, device=dev)
        _lib.call("hriemo_gate_dy_f32", _ops._p(dH), _ops._p(w), is_a, _ops._p(dpool), _ops._p(kpm), _ops._p(dY), B, L, Lx, d, st)
        dx, _, dgam, dbet, _ = add_ln_bwd(dY, x32, None, gamma, want_dbias=False)
        outs.append((dx.view(B, Lx, d), dgam, dbet))
    (dxa, dga, dba), (dxt, dgt, dbt) = outs
    twin_a, twin_t, dt_a, dt_t = ctx.f32_twins
    ga16 = None if twin_a else (dxa.to(dt_a) if ctx.needs_input_grad[0] else None)
    ga32 = dxa if (twin_a and ctx.needs_input_grad[1]) else None
    gt16 = None if twin_t else (dxt.to(dt_t) if ctx.needs_input_grad[2] else None)
    gt32 = dxt if (twin_t and ctx.needs_input_grad[3]) else None
    return ga16, ga32, gt16, gt32, dga, dba, dgt, dbt, dw1, db1, dw2, db2, None, None, None


def linear_any_k(ctx, x, w, b, sh):
    """LinearFn (MOSEI projections, K = 74 / 300): contraction padded to a multiple of 8"""
    _ops._require_fp32_masters(w, b)
    _ops._require_gpu(x)
    K = x.shape[-1]
    N = w.shape[0]
    M = x.numel() // K
    kp = (K + 7) // 8 * 8
    xp = torch.zeros((M, kp), dtype=F32, device=x.device)
    xp[:, :K].copy_(x.reshape(M, K))
    if N % 8:
        raise ValueError(f"fp32 mode: output width {N} must be a multiple of 8")
    y = linear(xp, sh, w, b, kp=kp)
    ctx.fp32 = True
    if recording(ctx):
        ctx.save_for_backward(xp)
        ctx.f32_cfg = (tuple(x.shape), x.dtype, K, N, M, kp)
        ctx.f32_params = (w, b, sh)
    return y.view(*x.shape[:-1], N)


def linear_any_k_bwd(ctx, dy):
    (xp,) = ctx.saved_tensors
    shape, xdtype, K, N, M, kp = ctx.f32_cfg
    w, b, sh = ctx.f32_params
    dy2 = _c(dy.float()).view(M, N)
    dx = None
    if ctx.needs_input_grad[0]:
        dx = linear_dx(dy2, sh, w, kp=kp)[:, :K].to(xdtype).reshape(shape)
    dw = linear_dw(dy2, xp)[:, :K].contiguous()
    return dx, dw, colsum(dy2), None


def expand_bwd(dout, d32, B, Ne, d, like):
    """queries[N_e,d] -> [B,N_e,d]: the gradient is the sum over the batch, in fp32"""
    g = _total(dout, d32, (B, Ne * d))
    if g is None:
        return torch.zeros((Ne, d), dtype=F32, device=like.device)
    return colsum(g).view(Ne, d)


def rowdot_bwd(dl, z32, wf, B, Ne, d):
    """logits = z . w + b: -> (dz32 [B,Ne,d], dw [1,d], db [1])"""
    M = B * Ne
    dl2 = _c(dl.float()).view(M)
    dz = _new((M, d), z32)
    dw = _new((1, d), z32)
    db = _new((1,), z32)
    _lib.call("hriemo_rowdot_bwd_f32", _ops._p(dl2), _ops._p(z32), _ops._p(wf), _ops._p(dz), _ops._p(dw), _ops._p(db), 0, M, d, _ops._stream())
    return dz.view(B, Ne, d), dw, db
