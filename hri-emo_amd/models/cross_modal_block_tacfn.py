"""Drop-in for the reference's models/cross_modal_block_tacfn.py (CrossModalBlock :6-125,
CrossModalTransformer :130-166): same constructor arguments, forward signature, return tuples and
state_dict keys; the arithmetic runs in the gfx950 kernels of libhriemo.so.

The torch.nn modules created here (MultiheadAttention, LayerNorm, Linear) are PARAMETER CONTAINERS only
-- identical key names and default initialisation as the reference, their forward() is never called."""
import torch
import torch.nn as nn

from .. import _ops


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

    def _self(self, x, mha, ln, kpm, p, seed, site, need_w):
        return _ops.SelfAttnLN.apply(x, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias,
                                     ln.weight, ln.bias, self._sh, self.n_heads, kpm, p, seed, site,
                                     self.batch_offset, need_w)

    def _cross(self, xq, xkv, mha, ln, kpm, p, seed, site, need_w):
        return _ops.CrossAttnLN.apply(xq, xkv, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight,
                                      mha.out_proj.bias, ln.weight, ln.bias, self._sh, self.n_heads, kpm, p, seed,
                                      site, self.batch_offset, need_w)

    def _ffn(self, x, ffn, ln, p, seed, site):
        return _ops.FFNLN.apply(x, ffn[0].weight, ffn[0].bias, ffn[2].weight, ffn[2].bias, ln.weight, ln.bias,
                                self._sh, p, 0.0, seed, site, self.batch_offset)

    def forward(self, h_a, h_t, mask_a=None, mask_t=None, return_attention=False):
        out_dtype = h_a.dtype          # outputs come back in the caller's dtype (bf16 inside)
        h_a, h_t = _ops.to_bf16(h_a), _ops.to_bf16(h_t)
        B, La, _ = h_a.shape
        Lt = h_t.shape[1]
        kpm_a, kpm_t = _ops.mask_u8(mask_a, B, La), _ops.mask_u8(mask_t, B, Lt)
        p = self.p if self.training else 0.0
        seed = _ops.next_seed(self.training and p > 0)
        s = self._site
        need = bool(return_attention)
        h_a_self, w_a = self._self(h_a, self.self_attn_a, self.self_norm_a, kpm_a, p, seed, s[0], need)      # :74-81
        h_t_self, w_t = self._self(h_t, self.self_attn_t, self.self_norm_t, kpm_t, p, seed, s[1], need)      # :85-92
        x, w_a2t = self._cross(h_a_self, h_t_self, self.attn_a2t, self.norm_a1, kpm_t, p, seed, s[2], need)  # :98-105
        h_a_cm = self._ffn(x, self.ffn_a, self.norm_a2, p, seed, s[3])                                       # :106
        x, w_t2a = self._cross(h_t_self, h_a_self, self.attn_t2a, self.norm_t1, kpm_a, p, seed, s[4], need)  # :111-118
        h_t_cm = self._ffn(x, self.ffn_t, self.norm_t2, p, seed, s[5])                                       # :119
        h_a_cm, h_t_cm = h_a_cm.to(out_dtype), h_t_cm.to(out_dtype)
        if return_attention:
            return h_a_cm, h_t_cm, {"audio_self": w_a, "text_self": w_t, "audio_queries_text": w_a2t,
                                    "text_queries_audio": w_t2a}
        return h_a_cm, h_t_cm


class CrossModalTransformer(nn.Module):
    def __init__(self, num_layers=2, d_model=768, n_heads=8, dropout=0.1):
        super().__init__()
        self.layers = nn.ModuleList([CrossModalBlock(d_model, n_heads, dropout) for _ in range(num_layers)])

    def forward(self, h_a, h_t, mask_a=None, mask_t=None, return_attention=False):
        all_layers_attn = []
        for layer in self.layers:
            if return_attention:
                h_a, h_t, maps = layer(h_a, h_t, mask_a, mask_t, return_attention=True)
                all_layers_attn.append(maps)
            else:
                h_a, h_t = layer(h_a, h_t, mask_a, mask_t, return_attention=False)
        if return_attention:
            return h_a, h_t, all_layers_attn
        return h_a, h_t
