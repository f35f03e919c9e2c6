"""hri_emo_amd -- MI355X-native (gfx950) implementation of HRI-EMO's cross-modal fusion + beta-gate +
emotion-decoder forward/backward path behind the reference's own ``models/*`` nn.Module API.

Import as ``hri_emo_amd`` (the repo-root shim ``hri_emo_amd.py`` maps the module name onto this
``hri-emo_amd/`` directory).  ``hri_emo_amd.models.<file>`` mirrors ``models/<file>`` of the reference.
"""
from . import _lib  # noqa: F401  (fails loudly if libhriemo.so is missing when first used)
from .models.cross_modal_block_tacfn import CrossModalBlock, CrossModalTransformer  # noqa: F401
from .models.beta_gate_tacfn import BetaGate  # noqa: F401
from .models.emotion_decoder import EmotionDecoder, ExplainableDecoderLayer  # noqa: F401
from .models.fusion_with_emotion_decoder import FusionWithEmotionDecoder  # noqa: F401
from .models.mosei_fusion_with_emotion_decoder import MoseiFusionWithEmotionDecoder  # noqa: F401
from .models.fusion_classifier import FusionClassifier  # noqa: F401

from ._ops import set_gemm_mode, gemm_mode, enable_fused_wgrad, set_precision, precision, set_varlen, varlen  # noqa: F401

__all__ = ["set_gemm_mode", "gemm_mode", "enable_fused_wgrad", "set_precision", "precision", "set_varlen", "varlen", "CrossModalBlock", "CrossModalTransformer", "BetaGate", "EmotionDecoder", "ExplainableDecoderLayer",
           "FusionWithEmotionDecoder", "MoseiFusionWithEmotionDecoder", "FusionClassifier"]
