"""Drop-in for the reference's legacy models/beta_gate.py (masked_mean :6-32, scalar BetaGate :35-114) -- the gate
the reference's own tests/test_beta_gate.py builds: no LayerNorm, one beta per sample."""
import torch.nn as nn

try:
    from .. import _ops
except ImportError:            # imported as top-level `models` (PYTHONPATH=<repo>/hri-emo_amd, the reference's import path)
    from hri_emo_amd import _ops


class BetaGate(nn.Module):
    def __init__(self, d_model=768, hidden_dim=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(d_model * 4, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, 1))
        self._sh = _ops.Shadows()

    def forward(self, h_a, h_t, mask_a=None, mask_t=None):
        out_dtype = h_a.dtype
        B, La, _ = h_a.shape
        Lt = h_t.shape[1]
        kpm_a, kpm_t = _ops.mask_u8(mask_a, B, La), _ops.mask_u8(mask_t, B, Lt)
        h_fusion, beta = _ops.LegacyBetaGateFn.apply(h_a, h_t, self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight,
                                                     self.mlp[2].bias, self._sh, kpm_a, kpm_t)
        return h_fusion.to(out_dtype), beta
