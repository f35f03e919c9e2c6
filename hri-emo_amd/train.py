"""Losses of the reference's trainers as ONE kernel each (value + gradients, hriemo_fusion_loss):
  fusion_step_loss        scripts/fusion/train_fusion_seq_level_decoder.py:312-326 (multi-label branch):
                          BCEWithLogits(logits, y) - 0.01*mean(beta*(1-beta))
  fusion_step_loss_single_label   the same trainer's single_label branch (:312-314, criterion nn.CrossEntropyLoss() :413-414):
                          CrossEntropy(logits, class index) - 0.01*mean(beta*(1-beta))   (hriemo_fusion_loss_ce)
  mosei_step_loss         scripts/fusion/train_mosei_fusion_seq_level_decoder.py:340-347,383-387,569:
                          BCEWithLogits(logits, y; pos_weight) + beta_entropy*H(beta), divided by grad_accum
CPU tensors (the gloo tests drive the CPU oracle through dp.py) take the same formulas in plain torch."""
import torch
import torch.nn.functional as F


def fusion_step_loss(logits, beta, targets):
    if logits.is_cuda:
        from ._ops import FusionLossFn
        return FusionLossFn.apply(logits, beta, targets, None, 1, 0.01, 1.0)
    return F.binary_cross_entropy_with_logits(logits, targets) - 0.01 * (beta * (1.0 - beta)).mean()


def fusion_step_loss_single_label(logits, beta, labels):
    if logits.is_cuda:
        from ._ops import FusionLossCEFn
        return FusionLossCEFn.apply(logits, beta, labels, 1, 0.01, 1.0)
    return F.cross_entropy(logits, labels) - 0.01 * (beta * (1.0 - beta)).mean()


def beta_entropy_loss(beta, eps=1e-8):
    b = torch.clamp(beta, eps, 1.0 - eps)
    return (-(b * torch.log(b) + (1 - b) * torch.log(1 - b))).mean()


def mosei_step_loss(logits, beta, targets, pos_weight=None, beta_entropy=0.0, grad_accum=1):
    if logits.is_cuda:
        from ._ops import FusionLossFn
        return FusionLossFn.apply(logits, beta if beta_entropy > 0 else None, targets, pos_weight, 2, float(beta_entropy), 1.0 / grad_accum)
    loss = F.binary_cross_entropy_with_logits(logits, targets, pos_weight=pos_weight)
    if beta is not None and beta_entropy > 0:
        loss = loss + beta_entropy * beta_entropy_loss(beta)
    return loss / grad_accum
