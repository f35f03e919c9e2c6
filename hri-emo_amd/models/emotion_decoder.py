"""Drop-in for the reference's models/emotion_decoder.py (ExplainableDecoderLayer :5-64, EmotionDecoder :66-162)."""
import torch
import torch.nn as nn

try:
    from .. import _ops
except ImportError:            # imported as top-level `models` (PYTHONPATH=<repo>/hri-emo_amd, the reference's import path)
    from hri_emo_amd import _ops


def _memory16(memory):
    """bf16 GEMM operand of the encoder memory; in the fp32 inference mode it carries the caller's fp32 values along"""
    m16 = _ops.to_bf16(memory)
    if _ops.precision() == "fp32" and memory.dtype != torch.bfloat16:
        from hri_emo_amd import _fp32
        m16 = _fp32.tag32(m16, memory.float())
    return m16


_HOIST_KV = True           # the memory's K | V projections of all layers issued up front on the side stream (EmotionDecoder._fwd)


class ExplainableDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.nhead, self.p = nhead, float(dropout)
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=True)
        self.norm1 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.cross_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=True)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout2 = nn.Dropout(dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.activation = nn.ReLU()
        self._sh = _ops.Shadows()
        self._site = [_ops.new_site_base() for _ in range(3)]
        self.batch_offset = 0

    def _self_block(self, tgt, tgt32):
        """the query self-attention sub-layer (:42-43); -> (tgt, tgt32, the layer's dropout seed)"""
        p = self.p if self.training else 0.0
        seed = _ops.next_seed(self.training and p > 0)
        sa = self.self_attn
        tgt, tgt32, _ = _ops.SelfAttnLN.apply(tgt, tgt32, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight,
                                              sa.out_proj.bias, self.norm1.weight, self.norm1.bias, self._sh,
                                              self.nhead, None, p, seed, self._site[0], self.batch_offset, False)     # :42-43
        return tgt, tgt32, seed

    def _fwd_pair(self, tgt, tgt32, memory, memory_key_padding_mask, need, kv_pre=None, kv_ready=None):
        B, L, _ = memory.shape
        kpm = _ops.mask_u8(memory_key_padding_mask, B, L)
        p = self.p if self.training else 0.0
        ca, s = self.cross_attn, self._site
        tgt, tgt32, seed = self._self_block(tgt, tgt32)
        if kv_ready is not None:          # K | V of the memory were projected on the side stream (EmotionDecoder._fwd)
            torch.cuda.current_stream(memory.device).wait_event(kv_ready)
        tgt, tgt32, w = _ops.CrossAttnLN.apply(tgt, tgt32, memory, ca.in_proj_weight, ca.in_proj_bias,
                                               ca.out_proj.weight, ca.out_proj.bias, self.norm2.weight,
                                               self.norm2.bias, self._sh, self.nhead, kpm, p, seed, s[1],
                                               self.batch_offset, need, kv_pre)                               # :48-55
        tgt, tgt32 = _ops.FFNLN.apply(tgt, tgt32, self.linear1.weight, self.linear1.bias, self.linear2.weight,
                                      self.linear2.bias, self.norm3.weight, self.norm3.bias, self._sh, p, p, seed,
                                      s[2], self.batch_offset)                                                # :58-59
        return tgt, tgt32, w

    def forward(self, tgt, memory, memory_key_padding_mask=None, return_attention=False):
        out_dtype = tgt.dtype
        t16, t32 = _ops.as_pair(tgt)
        mem = _memory16(memory)
        if _ops.precision() == "fp32":
            from hri_emo_amd import _fp32
            mem = _fp32.f32_of(mem)
        t16, t32, w = self._fwd_pair(t16, t32, mem, memory_key_padding_mask, bool(return_attention))
        tgt = _ops.from_pair(t16, t32, out_dtype)
        return (tgt, w) if return_attention else (tgt, None)


class EmotionDecoder(nn.Module):
    def __init__(self, d_model: int = 768, num_emotions: int = 4, n_heads: int = 8, num_layers: int = 2,
                 dim_feedforward: int = 2048, dropout: float = 0.1, use_output_layer: bool = True):
        super().__init__()
        self.d_model, self.num_emotions, self.use_output_layer = d_model, num_emotions, use_output_layer
        self.emotion_queries = nn.Parameter(torch.randn(num_emotions, d_model))
        self.layers = nn.ModuleList([ExplainableDecoderLayer(d_model, n_heads, dim_feedforward, dropout)
                                     for _ in range(num_layers)])
        self.out_proj = nn.Linear(d_model, 1) if use_output_layer else None

    def _queries(self, B):
        # :127; the fp32 twin of the broadcast queries (residual operand of the first layer; gradient flows via `out`) comes from
        # the same launch
        if _ops.TWIN and self.emotion_queries.is_cuda:          # (fp32 mode: the fp32 copy carries the gradient back)
            return _ops.ExpandFn.apply(self.emotion_queries, B, True)
        out = _ops.ExpandFn.apply(self.emotion_queries, B)
        out32 = self.emotion_queries.detach().float().unsqueeze(0).expand(B, -1, -1).contiguous() if _ops.TWIN else None
        return out, out32

    def _fwd(self, memory16, memory_key_padding_mask, need, out_dtype):
        B = memory16.size(0)
        out, out32 = self._queries(B)
        all_layers_attn = []
        if _ops.want_mx_copy(memory16.shape[0] * memory16.shape[1], memory16.shape[2]):
            # fp8 GEMM mode: every layer projects the same memory to K | V -- quantise it once
            m2 = memory16 if memory16.is_contiguous() else memory16.contiguous()
            memory16 = _ops.tag_mx(m2, _ops.quant_mx8(m2.view(-1, m2.shape[2])))
        if _ops.precision() == "fp32":
            from hri_emo_amd import _fp32
            memory16 = _fp32.f32_of(memory16)          # the layers read the fp32 values of the memory (h_fusion's twin)
        # The memory's K | V projections (one [B*L_f, 2d] GEMM per layer, the only chip-sized launches of the decoder) do not depend
        # on the queries: all layers' are issued up front on the side stream and run -- forward and, through autograd, backward --
        # beside the decoder's serial chain of M = B*N_e launches instead of inside it.
        kvs, ready = [None] * len(self.layers), None
        side = _ops.side_stream(memory16.device) if (_HOIST_KV and _ops.precision() == "bf16" and memory16.is_cuda) else None
        if side is not None:
            main = torch.cuda.current_stream(memory16.device)
            _ops.fork(side, main)
            _ops.share(memory16, side)
            with torch.cuda.stream(side):
                jm = _ops.grad_join(len(self.layers), always=True) if (len(self.layers) > 1 and memory16.requires_grad) else None      # the layers' memory gradients meet in one dX GEMM (only if the memory needs one: frozen layers would strand a deposit)
                kvs = [_ops.KVProjFn.apply(memory16, l.cross_attn.in_proj_weight, l.cross_attn.in_proj_bias, l._sh, jm) for l in self.layers]
                ready = torch.cuda.Event()
                ready.record(side)
            for kv in kvs:
                _ops.share(kv, main)
        for i, layer in enumerate(self.layers):
            out, out32, attn_map = layer._fwd_pair(out, out32, memory16, memory_key_padding_mask, need, kvs[i],
                                                   ready if i == 0 else None)
            if need and attn_map is not None:
                all_layers_attn.append(attn_map)
        logits = None
        if self.out_proj is not None:
            logits = _ops.RowDotFn.apply(out, out32, self.out_proj.weight, self.out_proj.bias)           # :155
        z = _ops.from_pair(out, out32, out_dtype)
        return z, logits, all_layers_attn

    def forward(self, memory, memory_key_padding_mask=None, return_attention=False):
        z, logits, maps = self._fwd(_memory16(memory), memory_key_padding_mask, bool(return_attention), memory.dtype)
        if return_attention:
            return z, logits, maps
        return z, logits
