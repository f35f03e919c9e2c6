"""Drop-in for the reference's models/fusion_with_emotion_decoder.py (FusionWithEmotionDecoder :10-197)."""
import torch
import torch.nn as nn

try:
    from .. import _ops
except ImportError:            # imported as top-level `models` (PYTHONPATH=<repo>/hri-emo_amd, the reference's import path)
    from hri_emo_amd import _ops
from .cross_modal_block_tacfn import CrossModalTransformer
from .beta_gate_tacfn import BetaGate
from .emotion_decoder import EmotionDecoder


class FusionWithEmotionDecoder(nn.Module):
    def __init__(self, d_model: int = 768, num_emotions: int = 4, n_heads: int = 8, num_layers_fusion: int = 2,
                 num_layers_decoder: int = 2, beta_hidden: int = 256, dropout: float = 0.1):
        super().__init__()
        self.cross_modal = CrossModalTransformer(num_layers=num_layers_fusion, d_model=d_model, n_heads=n_heads,
                                                 dropout=dropout)
        self.beta_gate = BetaGate(d_model=d_model, hidden_dim=beta_hidden)
        self.emotion_decoder = EmotionDecoder(d_model=d_model, num_emotions=num_emotions, n_heads=n_heads,
                                              num_layers=num_layers_decoder, dropout=dropout, use_output_layer=True)

    def set_batch_offset(self, offset: int):
        """Global index of this shard's first utterance: keeps dropout masks independent of how the
        batch is sharded across GPUs (hri_emo_amd.dp)."""
        for m in self.modules():
            if hasattr(m, "batch_offset"):
                m.batch_offset = int(offset)

    def _prefetch_shadows(self, device):
        """bf16 copies of the gate's and the decoder's weight matrices, cast on the side stream at the top of the step, beside the
        first fusion layer, instead of one ~5 us launch in front of every GEMM of the decoder's latency-bound chain.  Returns the
        event the consumer stream waits for, or None (one stream / fp32 / fp8 mode)."""
        if _ops.precision() != "bf16" or not device.type == "cuda":
            return None
        side = _ops.side_stream(device)
        if side is None:
            return None
        main = torch.cuda.current_stream(device)
        _ops.fork(side, main)                       # the masters may have just been updated on the caller's stream
        with torch.cuda.stream(side):
            g = self.beta_gate
            pairs = [(g._sh, g.mlp[0].weight), (g._sh, g.mlp[2].weight)]
            for layer in self.emotion_decoder.layers:
                pairs += [(layer._sh, w) for w in (layer.self_attn.in_proj_weight, layer.self_attn.out_proj.weight,
                                                   layer.cross_attn.in_proj_weight, layer.cross_attn.out_proj.weight,
                                                   layer.linear1.weight, layer.linear2.weight)]
            _ops.prefetch_batch(pairs)          # one launch for the fourteen matrices (the step time is the same as with fourteen:
            #                                     7.45 vs 7.45 ms -- the graph runtime starts the branches late either way)
            ev = torch.cuda.Event()
            ev.record(side)
        return ev

    def _ensure_3d(self, x):
        if x.dim() == 2:
            return x.unsqueeze(1)
        if x.dim() == 3:
            return x
        raise ValueError(f"Expected 2D or 3D tensor, got {x.shape}")

    def _build_fused_mask(self, mask_a, mask_t, L_fused):
        if mask_a is None and mask_t is None:
            return None

        def fit(m):
            if m is None:
                return None
            if m.size(1) < L_fused:
                pad = torch.ones(m.size(0), L_fused - m.size(1), dtype=torch.bool, device=m.device)
                return torch.cat([m, pad], dim=1)
            return m[:, :L_fused]

        ma, mt = fit(mask_a), fit(mask_t)
        if ma is None:
            return mt
        if mt is None:
            return ma
        return ma | mt

    def forward(self, h_a, h_t, mask_a=None, mask_t=None, return_attention=False):
        h_a, h_t = self._ensure_3d(h_a), self._ensure_3d(h_t)
        out_dtype = h_a.dtype
        need = bool(return_attention)
        # one cast at the boundary; between the sub-modules activations travel as (bf16, fp32-twin) pairs
        a, a32 = _ops.as_pair(h_a)
        t, t32 = _ops.as_pair(h_t)
        _ops.begin_step()
        # the decoder's memory mask needs the two padding masks only (L_fused = T_t, beta_gate_tacfn.py:98-116): built here, not on
        # the decoder's serial chain behind the gate
        fused_early = self._build_fused_mask(mask_a, mask_t, h_t.size(1))
        ready = self._prefetch_shadows(a.device)
        _ops.CTX.join_scope += 1          # logits, beta and z all depend on both branches: the encoder's gradient joins are safe
        try:
            a, a32, t, t32, encoder_attns = self.cross_modal._fwd_pair(a, a32, t, t32, mask_a, mask_t, need)
        finally:
            _ops.CTX.join_scope -= 1
        if ready is not None:
            torch.cuda.current_stream(a.device).wait_event(ready)
        h_fusion, beta = self.beta_gate._fwd_pair(a, a32, t, t32, mask_a, mask_t)
        fused_mask = fused_early if h_fusion.size(1) == h_t.size(1) else self._build_fused_mask(mask_a, mask_t, h_fusion.size(1))
        z, logits, decoder_attns = self.emotion_decoder._fwd(h_fusion, fused_mask, need, out_dtype)
        if return_attention:
            return logits, beta, z, {"encoder": encoder_attns, "decoder": decoder_attns}
        return logits, beta, z
