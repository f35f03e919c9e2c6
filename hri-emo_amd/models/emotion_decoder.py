"""Drop-in for the reference's models/emotion_decoder.py (ExplainableDecoderLayer :5-64, EmotionDecoder :66-162)."""
import torch
import torch.nn as nn

from .. import _ops


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

    def forward(self, tgt, memory, memory_key_padding_mask=None, return_attention=False):
        out_dtype = tgt.dtype
        tgt, memory = _ops.to_bf16(tgt), _ops.to_bf16(memory)
        B, L, _ = memory.shape
        kpm = _ops.mask_u8(memory_key_padding_mask, B, L)
        p = self.p if self.training else 0.0
        seed = _ops.next_seed(self.training and p > 0)
        sa, ca, s = self.self_attn, self.cross_attn, self._site
        tgt, _ = _ops.SelfAttnLN.apply(tgt, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias,
                                       self.norm1.weight, self.norm1.bias, self._sh, self.nhead, None, p, seed, s[0],
                                       self.batch_offset, False)                                          # :42-43
        tgt, w = _ops.CrossAttnLN.apply(tgt, memory, ca.in_proj_weight, ca.in_proj_bias, ca.out_proj.weight,
                                        ca.out_proj.bias, self.norm2.weight, self.norm2.bias, self._sh, self.nhead,
                                        kpm, p, seed, s[1], self.batch_offset, bool(return_attention))   # :48-55
        tgt = _ops.FFNLN.apply(tgt, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                               self.norm3.weight, self.norm3.bias, self._sh, p, p, seed, s[2], self.batch_offset)  # :58-59
        tgt = tgt.to(out_dtype)
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

    def forward(self, memory, memory_key_padding_mask=None, return_attention=False):
        out_dtype = memory.dtype
        memory = _ops.to_bf16(memory)
        B = memory.size(0)
        out = _ops.ExpandFn.apply(self.emotion_queries, B)                                                # :127
        all_layers_attn = []
        for layer in self.layers:
            out, attn_map = layer(out, memory, memory_key_padding_mask, return_attention)
            if return_attention and attn_map is not None:
                all_layers_attn.append(attn_map)
        logits = None
        if self.out_proj is not None:
            logits = _ops.RowDotFn.apply(out, self.out_proj.weight, self.out_proj.bias)                  # :155
        z = out.to(out_dtype)
        if return_attention:
            return z, logits, all_layers_attn
        return z, logits
