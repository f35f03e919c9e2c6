"""Drop-in for the reference's models/cross_modal_block_tacfn.py (CrossModalBlock :6-125,
CrossModalTransformer :130-166): same constructor arguments, forward signature, return tuples and
state_dict keys; the arithmetic runs in the gfx950 kernels of libhriemo.so.

The torch.nn modules created here (MultiheadAttention, LayerNorm, Linear) are PARAMETER CONTAINERS only
-- identical key names and default initialisation as the reference, their forward() is never called.

Internally every activation of the residual stream travels as a pair (bf16 copy for the GEMM operands,
fp32 twin for the residual/LayerNorm path): see _ops.as_pair and DESIGN.md (precision)."""
import torch
import torch.nn as nn

try:
    from .. import _ops
except ImportError:            # imported as top-level `models` (PYTHONPATH=<repo>/hri-emo_amd, the reference's import path)
    from hri_emo_amd import _ops



class CrossModalBlock(nn.Module):
    def __init__(self, d_model=768, n_heads=8, dropout=0.1):
        super().__init__()
        self.d_model, self.n_heads, self.p = d_model, n_heads, float(dropout)
        self.self_attn_a = nn.MultiheadAttention(d_model, n_heads, dropout=dropout, batch_first=True)
        self.self_attn_t = nn.MultiheadAttention(d_model, n_heads, dropout=dropout, batch_first=True)
        self.self_norm_a = nn.LayerNorm(d_model)
        self.self_norm_t = nn.LayerNorm(d_model)
        self.attn_a2t = nn.MultiheadAttention(d_model, n_heads, dropout=dropout, batch_first=True)
        self.attn_t2a = nn.MultiheadAttention(d_model, n_heads, dropout=dropout, batch_first=True)
        self.ffn_a = nn.Sequential(nn.Linear(d_model, 4 * d_model), nn.ReLU(), nn.Linear(4 * d_model, d_model))
        self.ffn_t = nn.Sequential(nn.Linear(d_model, 4 * d_model), nn.ReLU(), nn.Linear(4 * d_model, d_model))
        self.norm_a1 = nn.LayerNorm(d_model)
        self.norm_a2 = nn.LayerNorm(d_model)
        self.norm_t1 = nn.LayerNorm(d_model)
        self.norm_t2 = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self._sh = _ops.Shadows()
        self._site = [_ops.new_site_base() for _ in range(6)]
        self.batch_offset = 0          # global index of this shard's first utterance (data parallel)

    def _self(self, x, x32, mha, ln, kpm, p, seed, site, need_w):
        return _ops.SelfAttnLN.apply(x, x32, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight,
                                     mha.out_proj.bias, ln.weight, ln.bias, self._sh, self.n_heads, kpm, p, seed, site,
                                     self.batch_offset, need_w)

    def _cross(self, xq, xq32, xkv, mha, ln, kpm, p, seed, site, need_w, kv_pre=None, join_q=None, q_pre=None, slots=None):
        return _ops.CrossAttnLN.apply(xq, xq32, xkv, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight,
                                      mha.out_proj.bias, ln.weight, ln.bias, self._sh, self.n_heads, kpm, p, seed,
                                      site, self.batch_offset, need_w, kv_pre, join_q, q_pre, slots)

    def _shared_proj(self, x, mha_q, mha_kv, join):
        """[Q of the cross-attention `x` queries | K, V of the one it serves] from ONE N = 3d GEMM (_ops.SharedProjFn);
        -> (q, kv, SharedGrad the two attention backwards write dQ and dK | dV into)"""
        sg = _ops.SharedGrad(x.shape[0] * x.shape[1], x.shape[2], x.device)
        q, kv = _ops.SharedProjFn.apply(x, mha_q.in_proj_weight, mha_q.in_proj_bias, mha_kv.in_proj_weight, mha_kv.in_proj_bias,
                                        self._sh, join, sg)
        return q, kv, sg

    def _kv(self, xkv, mha, join):
        """K | V projection of a cross-attention as its own node (_ops.KVProjFn): the gradient it returns for `xkv` meets the
        gradient the OTHER cross-attention returns for the same tensor as its query side in a GradJoin"""
        return _ops.KVProjFn.apply(xkv, mha.in_proj_weight, mha.in_proj_bias, self._sh, join)

    def _ffn(self, x, x32, ffn, ln, p, seed, site, seq=None):
        return _ops.FFNLN.apply(x, x32, ffn[0].weight, ffn[0].bias, ffn[2].weight, ffn[2].bias, ln.weight, ln.bias,
                                self._sh, p, 0.0, seed, site, self.batch_offset, seq)

    def _fwd_pair(self, a, a32, t, t32, mask_a, mask_t, need, plan=None):
        """(bf16, fp32-twin) pairs in and out; returns (a, a32, t, t32, maps|None).
        plan = (Seq audio, Seq text): a / t hold the packed valid rows ([1, N, d], _ops.pack_pair) and the attention kernels get
        cu_seqlens instead of padding masks."""
        B, La, _ = a.shape
        Lt = t.shape[1]
        if plan is not None:
            kpm_a, kpm_t = plan                 # self-attention: the Seq itself; cross-attention: (query side, key side)
            kpm_a2t, kpm_t2a = (plan[0], plan[1]), (plan[1], plan[0])
        else:
            kpm_a, kpm_t = _ops.mask_u8(mask_a, B, La), _ops.mask_u8(mask_t, B, Lt)
            kpm_a2t, kpm_t2a = kpm_t, kpm_a
        p = self.p if self.training else 0.0
        seed = _ops.next_seed(self.training and p > 0)
        s = self._site
        _ops._require_gpu(a)
        # fp32 inference mode: the key/value side of a cross-attention reads the fp32 twin of the other branch, not its bf16 copy
        fp32 = _ops.precision() == "fp32"
        kv = lambda x16, x32: x32 if (fp32 and x32 is not None) else x16          # noqa: E731
        main = torch.cuda.current_stream(a.device)
        _ops.note_main_stream(main)
        side = _ops.side_stream(a.device)
        # Each self-attention output has two consumers -- the queries of its own cross-attention and the keys / values of the
        # other one.  With the K | V projections as their own nodes, created BEFORE both cross-attention cores, the engine runs
        # the cores' backward first (they deposit the query-side gradients) and the projections' backward last, where one
        # dX GEMM adds the deposit in its epilogue: no elementwise add launches on [B*L, d] (_ops.GradJoin).
        # A join exists only where BOTH consumers are certain to get a backward node that must produce the shared activation's
        # gradient, i.e. where that activation requires grad: with a frozen cross-attention and inputs that need no gradient the
        # partner's node never runs and a deposit would be stranded (autograd then sums whatever gradients there are).
        ja = jt = None
        use_kv = not fp32
        join_for = lambda x: _ops.grad_join(2) if (use_kv and x.requires_grad) else None          # noqa: E731
        d = a.shape[2]
        shared = use_kv and _ops.shared_proj()
        if shared and side is None:
            a_s, a_s32, w_a = self._self(a, a32, self.self_attn_a, self.self_norm_a, kpm_a, p, seed, s[0], need)   # :74-81
            t_s, t_s32, w_t = self._self(t, t32, self.self_attn_t, self.self_norm_t, kpm_t, p, seed, s[1], need)   # :85-92
            ja, jt = join_for(a_s), join_for(t_s)
            q_a2t, kv_t2a, sga = self._shared_proj(a_s, self.attn_a2t, self.attn_t2a, ja)
            q_t2a, kv_a2t, sgt = self._shared_proj(t_s, self.attn_t2a, self.attn_a2t, jt)
            x, x32, w_a2t = self._cross(a_s, a_s32, t_s, self.attn_a2t, self.norm_a1, kpm_a2t, p, seed, s[2], need,
                                        kv_a2t, ja, q_a2t, (sga.slot(0, d), sgt.slot(d, 3 * d)))                   # :98-105
            a_cm, a_cm32 = self._ffn(x, x32, self.ffn_a, self.norm_a2, p, seed, s[3], plan[0] if plan is not None else None)
            x, x32, w_t2a = self._cross(t_s, t_s32, a_s, self.attn_t2a, self.norm_t1, kpm_t2a, p, seed, s[4], need,
                                        kv_t2a, jt, q_t2a, (sgt.slot(0, d), sga.slot(d, 3 * d)))                   # :111-118
            t_cm, t_cm32 = self._ffn(x, x32, self.ffn_t, self.norm_t2, p, seed, s[5], plan[1] if plan is not None else None)
        elif shared:
            # two streams, one GEMM per shared input: each branch projects its own self-attention output to [Q | K, V] on its own
            # stream (no dependence on the other branch yet), THEN the branches exchange the K | V halves and run the cores
            # (the audio branch -- three times the rows, the critical path -- is ENQUEUED first at every fork: the order of capture
            # decides which branch the graph runtime starts first)
            _ops.fork(side, main)
            for x_ in (t, t32, kpm_t, kpm_a):
                _ops.share(x_, side)

            def text_self():
                with torch.cuda.stream(side):
                    t_s_, t_s32_, w_t_ = self._self(t, t32, self.self_attn_t, self.self_norm_t, kpm_t, p, seed, s[1], need)
                    jt_ = join_for(t_s_)
                    return (t_s_, t_s32_, w_t_, jt_) + self._shared_proj(t_s_, self.attn_t2a, self.attn_a2t, jt_)

            a_s, a_s32, w_a = self._self(a, a32, self.self_attn_a, self.self_norm_a, kpm_a, p, seed, s[0], need)
            ja = join_for(a_s)
            q_a2t, kv_t2a, sga = self._shared_proj(a_s, self.attn_a2t, self.attn_t2a, ja)
            t_s, t_s32, w_t, jt, q_t2a, kv_a2t, sgt = text_self()
            main.wait_stream(side)
            _ops.fork(side, main)
            _ops.share(kv_a2t, main)
            _ops.share(kv_t2a, side)

            def text_cross():
                with torch.cuda.stream(side):
                    x_, x32_, w_ = self._cross(t_s, t_s32, a_s, self.attn_t2a, self.norm_t1, kpm_t2a, p, seed, s[4], need,
                                               kv_t2a, jt, q_t2a, (sgt.slot(0, d), sga.slot(d, 3 * d)))
                    return self._ffn(x_, x32_, self.ffn_t, self.norm_t2, p, seed, s[5], plan[1] if plan is not None else None) + (w_,)

            x, x32, w_a2t = self._cross(a_s, a_s32, t_s, self.attn_a2t, self.norm_a1, kpm_a2t, p, seed, s[2], need,
                                        kv_a2t, ja, q_a2t, (sga.slot(0, d), sgt.slot(d, 3 * d)))
            a_cm, a_cm32 = self._ffn(x, x32, self.ffn_a, self.norm_a2, p, seed, s[3], plan[0] if plan is not None else None)
            t_cm, t_cm32, w_t2a = text_cross()
            main.wait_stream(side)
            for x_ in (t_cm, t_cm32, w_t, w_t2a):
                _ops.share(x_, main)
        elif side is None:
            a_s, a_s32, w_a = self._self(a, a32, self.self_attn_a, self.self_norm_a, kpm_a, p, seed, s[0], need)   # :74-81
            t_s, t_s32, w_t = self._self(t, t32, self.self_attn_t, self.self_norm_t, kpm_t, p, seed, s[1], need)   # :85-92
            ja, jt = join_for(a_s), join_for(t_s)
            kv_t2a = self._kv(a_s, self.attn_t2a, ja) if use_kv else None
            kv_a2t = self._kv(t_s, self.attn_a2t, jt) if use_kv else None
            x, x32, w_a2t = self._cross(a_s, a_s32, kv(t_s, t_s32), self.attn_a2t, self.norm_a1, kpm_a2t, p, seed, s[2], need,
                                        kv_a2t, ja)                                                                # :98-105
            a_cm, a_cm32 = self._ffn(x, x32, self.ffn_a, self.norm_a2, p, seed, s[3], plan[0] if plan is not None else None)                              # :106
            x, x32, w_t2a = self._cross(t_s, t_s32, kv(a_s, a_s32), self.attn_t2a, self.norm_t1, kpm_t2a, p, seed, s[4], need,
                                        kv_t2a, jt)                                                                # :111-118
            t_cm, t_cm32 = self._ffn(x, x32, self.ffn_t, self.norm_t2, p, seed, s[5], plan[1] if plan is not None else None)                              # :119
        else:
            # The audio and text branches only meet at the two cross-attentions (each reads the OTHER branch's
            # self-attention output), so the text branch runs on a second stream: its small grids (B*T_t rows)
            # fill the CUs the audio branch leaves idle.  Fork/join with stream waits; tensors that cross
            # streams are recorded on the consumer stream (allocator safety); autograd replays the same streams
            # in backward.
            _ops.fork(side, main)
            for x_ in (t, t32, kpm_t, kpm_a):
                _ops.share(x_, side)
            # (audio first at every fork, the reference's order of sub-layers: cross_modal_block_tacfn.py:74-119)
            a_s, a_s32, w_a = self._self(a, a32, self.self_attn_a, self.self_norm_a, kpm_a, p, seed, s[0], need)
            with torch.cuda.stream(side):
                t_s, t_s32, w_t = self._self(t, t32, self.self_attn_t, self.self_norm_t, kpm_t, p, seed, s[1], need)
            main.wait_stream(side)
            _ops.fork(side, main)
            _ops.share(t_s, main)
            _ops.share(a_s, side)
            if fp32:
                _ops.share(t_s32, main)
                _ops.share(a_s32, side)
            kv_t2a = kv_a2t = None
            ja, jt = join_for(a_s), join_for(t_s)
            if use_kv:
                with torch.cuda.stream(side):
                    kv_t2a = self._kv(a_s, self.attn_t2a, ja)
                kv_a2t = self._kv(t_s, self.attn_a2t, jt)
            x, x32, w_a2t = self._cross(a_s, a_s32, kv(t_s, t_s32), self.attn_a2t, self.norm_a1, kpm_a2t, p, seed, s[2], need,
                                        kv_a2t, ja)
            a_cm, a_cm32 = self._ffn(x, x32, self.ffn_a, self.norm_a2, p, seed, s[3], plan[0] if plan is not None else None)
            with torch.cuda.stream(side):
                x, x32, w_t2a = self._cross(t_s, t_s32, kv(a_s, a_s32), self.attn_t2a, self.norm_t1, kpm_t2a, p, seed, s[4], need,
                                            kv_t2a, jt)
                t_cm, t_cm32 = self._ffn(x, x32, self.ffn_t, self.norm_t2, p, seed, s[5], plan[1] if plan is not None else None)
            main.wait_stream(side)
            for x_ in (t_cm, t_cm32, w_t, w_t2a):
                _ops.share(x_, main)
        maps = {"audio_self": w_a, "text_self": w_t, "audio_queries_text": w_a2t, "text_queries_audio": w_t2a} if need else None
        return a_cm, a_cm32, t_cm, t_cm32, maps

    def forward(self, h_a, h_t, mask_a=None, mask_t=None, return_attention=False):
        out_dtype = h_a.dtype          # outputs come back in the caller's dtype
        a, a32 = _ops.as_pair(h_a)
        t, t32 = _ops.as_pair(h_t)
        a, a32, t, t32, maps = self._fwd_pair(a, a32, t, t32, mask_a, mask_t, bool(return_attention))
        h_a_cm, h_t_cm = _ops.from_pair(a, a32, out_dtype), _ops.from_pair(t, t32, out_dtype)
        if return_attention:
            return h_a_cm, h_t_cm, maps
        return h_a_cm, h_t_cm


class CrossModalTransformer(nn.Module):
    def __init__(self, num_layers=2, d_model=768, n_heads=8, dropout=0.1):
        super().__init__()
        self.layers = nn.ModuleList([CrossModalBlock(d_model, n_heads, dropout) for _ in range(num_layers)])

    def _fwd_pair(self, a, a32, t, t32, mask_a, mask_t, need):
        all_layers_attn = []
        plan = None
        _ops.FLUSH_SITES.add(self.layers[0]._site[1])        # layer-0 text self-attention: the last text-branch backward (_ops._DeferredWgrad)
        if _ops.varlen() and not need and mask_a is not None and mask_t is not None and _ops.precision() == "bf16":
            # the encoder on the valid rows only (prefix masks, as the collate builds them); anything else
            # takes the padded path.  dp.DataParallelStep injects bucketed plans whose lengths are device data (_ops.CTX.seq_override).
            if _ops.CTX.seq_override is not None:
                sa, st = _ops.CTX.seq_override
            else:
                sa = _ops.seq_plan(mask_a, a.shape[0], a.shape[1])
                st = _ops.seq_plan(mask_t, t.shape[0], t.shape[1])
            if sa is not None and st is not None:
                plan = (sa, st)
                (a, a32), (t, t32) = _ops.pack_pair(a, a32, sa), _ops.pack_pair(t, t32, st)
        for i, layer in enumerate(self.layers):
            a, a32, t, t32, maps = layer._fwd_pair(a, a32, t, t32, mask_a, mask_t, need, plan)
            if need:
                all_layers_attn.append(maps)
        if plan is not None:
            (a, a32), (t, t32) = _ops.unpack_pair(a, a32, plan[0]), _ops.unpack_pair(t, t32, plan[1])
        return a, a32, t, t32, all_layers_attn

    def forward(self, h_a, h_t, mask_a=None, mask_t=None, return_attention=False):
        out_dtype = h_a.dtype
        a, a32 = _ops.as_pair(h_a)
        t, t32 = _ops.as_pair(h_t)
        a, a32, t, t32, maps = self._fwd_pair(a, a32, t, t32, mask_a, mask_t, bool(return_attention))
        h_a, h_t = _ops.from_pair(a, a32, out_dtype), _ops.from_pair(t, t32, out_dtype)
        if return_attention:
            return h_a, h_t, maps
        return h_a, h_t
