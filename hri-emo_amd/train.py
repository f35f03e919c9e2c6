"""Loss of the reference's seq-level trainer, multi-label branch
(scripts/fusion/train_fusion_seq_level_decoder.py:312-326): BCEWithLogits(logits, y) - 0.01*mean(beta*(1-beta)).
[B, N_e]-sized tensors: plain torch ops (plumbing, off the hot path)."""
import torch.nn.functional as F


def fusion_step_loss(logits, beta, targets):
    return F.binary_cross_entropy_with_logits(logits, targets) - 0.01 * (beta * (1.0 - beta)).mean()
