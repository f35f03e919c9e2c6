"""Drop-in for the reference's models/beta_gate_tacfn.py (masked_mean :6-24, BetaGate :27-118)."""
import torch
import torch.nn as nn

try:
    from .. import _ops
except ImportError:            # imported as top-level `models` (PYTHONPATH=<repo>/hri-emo_amd, the reference's import path)
    from hri_emo_amd import _ops


class BetaGate(nn.Module):
    def __init__(self, d_model: int = 768, hidden_dim: int = 256):
        super().__init__()
        self.d_model = d_model
        self.norm_a = nn.LayerNorm(d_model)
        self.norm_t = nn.LayerNorm(d_model)
        self.mlp = nn.Sequential(nn.Linear(d_model * 4, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, d_model))
        self._sh = _ops.Shadows()

    def _fwd_pair(self, a, a32, t, t32, mask_a, mask_t):
        """h_fusion comes back bf16 only (fp32 mode: fp32): it is consumed as a GEMM operand (the decoder's memory)."""
        B, La, _ = a.shape
        Lt = t.shape[1]
        kpm_a, kpm_t = _ops.mask_u8(mask_a, B, La), _ops.mask_u8(mask_t, B, Lt)
        # (fp32 mode: the same Function; h_fusion then IS the fp32 tensor, read as such by the decoder and seen by autograd)
        return _ops.BetaGateFn.apply(a, a32, t, t32, self.norm_a.weight, self.norm_a.bias, self.norm_t.weight,
                                     self.norm_t.bias, self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight,
                                     self.mlp[2].bias, self._sh, kpm_a, kpm_t)

    def forward(self, h_a, h_t, mask_a=None, mask_t=None):
        out_dtype = h_a.dtype
        a, a32 = _ops.as_pair(h_a)
        t, t32 = _ops.as_pair(h_t)
        h_fusion, beta = self._fwd_pair(a, a32, t, t32, mask_a, mask_t)
        return h_fusion.to(out_dtype), beta
