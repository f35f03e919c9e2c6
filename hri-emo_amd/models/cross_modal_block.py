"""Drop-in for the reference's LEGACY models/cross_modal_block.py (CrossModalBlock :5-62, CrossModalTransformer
:65-95): bidirectional cross-attention + FFN without the intra-modal self-attention; text->audio attention reads
the ORIGINAL h_a.  A strict subset of the TACFN block's kernels; it is what the
reference's tests/test_cross_modal_block.py and tests/test_beta_gate.py exercise."""
import torch
import torch.nn as nn

try:
    from .. import _ops
except ImportError:
    from hri_emo_amd import _ops


class CrossModalBlock(nn.Module):
    def __init__(self, d_model=768, n_heads=8, dropout=0.1):
        super().__init__()
        self.n_heads, self.p = n_heads, float(dropout)
        self.attn_a2t = nn.MultiheadAttention(embed_dim=d_model, num_heads=n_heads, dropout=dropout, batch_first=True)
        self.attn_t2a = nn.MultiheadAttention(embed_dim=d_model, num_heads=n_heads, dropout=dropout, batch_first=True)
        self.ffn_a = nn.Sequential(nn.Linear(d_model, 4 * d_model), nn.ReLU(), nn.Linear(4 * d_model, d_model))
        self.ffn_t = nn.Sequential(nn.Linear(d_model, 4 * d_model), nn.ReLU(), nn.Linear(4 * d_model, d_model))
        self.norm_a1 = nn.LayerNorm(d_model)
        self.norm_a2 = nn.LayerNorm(d_model)
        self.norm_t1 = nn.LayerNorm(d_model)
        self.norm_t2 = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self._sh = _ops.Shadows()
        self._site = [_ops.new_site_base() for _ in range(4)]
        self.batch_offset = 0

    def _fwd_pair(self, a, a32, t, t32, mask_a, mask_t):
        _ops._require_gpu(a)
        B, La, _ = a.shape
        Lt = t.shape[1]
        kpm_a, kpm_t = _ops.mask_u8(mask_a, B, La), _ops.mask_u8(mask_t, B, Lt)
        p = self.p if self.training else 0.0
        seed = _ops.next_seed(self.training and p > 0)
        s, H, sh, bo = self._site, self.n_heads, self._sh, self.batch_offset

        def cross(xq, xq32, xkv, mha, ln, kpm, site):
            y, y32, _ = _ops.CrossAttnLN.apply(xq, xq32, xkv, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight,
                                               mha.out_proj.bias, ln.weight, ln.bias, sh, H, kpm, p, seed, site, bo, False)
            return y, y32

        def ffn(x, x32, f, ln, site):
            return _ops.FFNLN.apply(x, x32, f[0].weight, f[0].bias, f[2].weight, f[2].bias, ln.weight, ln.bias, sh, p,
                                    0.0, seed, site, bo)

        x, x32 = cross(a, a32, t, self.attn_a2t, self.norm_a1, kpm_t, s[0])      # :46-50
        a_o, a_o32 = ffn(x, x32, self.ffn_a, self.norm_a2, s[1])                 # :51
        x, x32 = cross(t, t32, a, self.attn_t2a, self.norm_t1, kpm_a, s[2])      # :53-57 (original h_a)
        t_o, t_o32 = ffn(x, x32, self.ffn_t, self.norm_t2, s[3])                 # :58
        return a_o, a_o32, t_o, t_o32

    def forward(self, h_a, h_t, mask_a=None, mask_t=None):
        out_dtype = h_a.dtype
        a, a32 = _ops.as_pair(h_a)
        t, t32 = _ops.as_pair(h_t)
        a, a32, t, t32 = self._fwd_pair(a, a32, t, t32, mask_a, mask_t)
        return _ops.from_pair(a, a32, out_dtype), _ops.from_pair(t, t32, out_dtype)


class CrossModalTransformer(nn.Module):
    def __init__(self, num_layers=2, d_model=768, n_heads=8, dropout=0.1):
        super().__init__()
        self.layers = nn.ModuleList([CrossModalBlock(d_model, n_heads, dropout) for _ in range(num_layers)])

    def forward(self, h_a, h_t, mask_a=None, mask_t=None):
        out_dtype = h_a.dtype
        a, a32 = _ops.as_pair(h_a)
        t, t32 = _ops.as_pair(h_t)
        for layer in self.layers:
            a, a32, t, t32 = layer._fwd_pair(a, a32, t, t32, mask_a, mask_t)
        return _ops.from_pair(a, a32, out_dtype), _ops.from_pair(t, t32, out_dtype)
