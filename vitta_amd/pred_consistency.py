"""Prediction-consistency loss across temporally augmented views.

Interface mirror of utils/pred_consistency_utils.py:15-31.  CUDA tensors go through the fused HIP
kernel (softmax per view, mean over views, L1 and the full gradient in one launch); CPU tensors
(host-only plumbing) use the equivalent torch expression.
"""
import torch


def compute_pred_consis(preds):
    """preds: (batch, n_views, n_class) logits -> sum_v sum_{b,k} |softmax_v - mean_v softmax| / n_views."""
    if preds.dim() != 3:
        raise ValueError("preds must be (batch_size, n_views, n_class)")
    if preds.is_cuda:
        from . import ops
        return ops.pred_consis(preds.float())
    p = torch.softmax(preds, dim=2)
    return (p - p.mean(dim=1, keepdim=True)).abs().sum() / preds.shape[1]
