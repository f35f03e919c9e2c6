"""Drop-in for the reference's models/mosei_fusion_with_emotion_decoder.py (MoseiFusionWithEmotionDecoder
:8-79): two input projections (COVAREP d=74, GloVe d=300 -> d_model) in front of the fusion backbone; same
constructor arguments, return tuples and state_dict keys (``audio_proj.*``, ``text_proj.*``, ``backbone.*``), so
the reference's MOSEI checkpoints load."""
import torch
import torch.nn as nn

try:
    from .. import _ops
except ImportError:            # imported as top-level `models` (PYTHONPATH=<repo>/hri-emo_amd, the reference's import path)
    from hri_emo_amd import _ops
from .fusion_with_emotion_decoder import FusionWithEmotionDecoder


class MoseiFusionWithEmotionDecoder(nn.Module):
    def __init__(self, d_audio: int, d_text: int, d_model: int = 256, num_emotions: int = 6, n_heads: int = 4,
                 num_layers_fusion: int = 2, num_layers_decoder: int = 2, beta_hidden: int = 128,
                 dropout: float = 0.2):
        super().__init__()
        self.audio_proj = nn.Linear(d_audio, d_model)      # parameter containers (reference keys / init)
        self.text_proj = nn.Linear(d_text, d_model)
        self.backbone = FusionWithEmotionDecoder(d_model=d_model, num_emotions=num_emotions, n_heads=n_heads,
                                                 num_layers_fusion=num_layers_fusion,
                                                 num_layers_decoder=num_layers_decoder, beta_hidden=beta_hidden,
                                                 dropout=dropout)
        self._sh = _ops.Shadows()

    def set_batch_offset(self, offset: int):
        self.backbone.set_batch_offset(offset)

    def forward(self, h_a, h_t, mask_a=None, mask_t=None, return_attention=False):
        h_a_proj = _ops.LinearFn.apply(h_a, self.audio_proj.weight, self.audio_proj.bias, self._sh)   # :63
        h_t_proj = _ops.LinearFn.apply(h_t, self.text_proj.weight, self.text_proj.bias, self._sh)     # :65
        return self.backbone(h_a_proj, h_t_proj, mask_a, mask_t, return_attention=return_attention)   # :68-79
