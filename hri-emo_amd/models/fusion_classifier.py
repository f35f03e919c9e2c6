"""Drop-in for the reference's models/fusion_classifier.py (FusionClassifier :9-150): TACFN cross-modal
transformer + vector beta-gate (both on the HIP kernels) + mean-pool + classifier head.

The head works on [B, d] vectors (LayerNorm, Linear(d,d), ReLU, Dropout, Linear(d, num_classes)) and is NOT part
of the north-star hot path (component 8); it stays plain torch.nn on the device."""
import torch
import torch.nn as nn

try:
    from .. import _ops
except ImportError:
    from hri_emo_amd import _ops
from .cross_modal_block_tacfn import CrossModalTransformer
from .beta_gate_tacfn import BetaGate


class FusionClassifier(nn.Module):
    def __init__(self, d_model: int = 768, num_classes: int = 4, n_heads: int = 8, num_layers: int = 2,
                 beta_hidden: int = 256, dropout: float = 0.2):
        super().__init__()
        self.cross_modal = CrossModalTransformer(num_layers=num_layers, d_model=d_model, n_heads=n_heads, dropout=dropout)
        self.beta_gate = BetaGate(d_model=d_model, hidden_dim=beta_hidden)
        self.classifier = nn.Sequential(nn.LayerNorm(d_model), nn.Linear(d_model, d_model), nn.ReLU(),
                                        nn.Dropout(dropout), nn.Linear(d_model, num_classes))

    def _ensure_3d(self, x):
        if x.dim() == 2:
            return x.unsqueeze(1)
        if x.dim() == 3:
            return x
        raise ValueError(f"Expected 2D or 3D tensor, got shape {x.shape}")

    def forward(self, h_a, h_t, mask_a=None, mask_t=None):
        h_a, h_t = self._ensure_3d(h_a), self._ensure_3d(h_t)
        a, a32 = _ops.as_pair(h_a)
        t, t32 = _ops.as_pair(h_t)
        a, a32, t, t32, _ = self.cross_modal._fwd_pair(a, a32, t, t32, mask_a, mask_t, False)      # :139
        h_fusion, beta = self.beta_gate._fwd_pair(a, a32, t, t32, mask_a, mask_t)                 # :142
        h_fusion_pooled = h_fusion.float().mean(dim=1)                                            # :145
        logits = self.classifier(h_fusion_pooled)                                                 # :148
        return logits, beta, h_fusion_pooled.to(h_a.dtype if h_a.dtype != torch.bfloat16 else torch.float32)
