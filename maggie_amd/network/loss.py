"""Loss helpers of the hot path that are not part of the fused per-scale pipeline (functional.matting_losses: weighted L1 + Laplacian
pyramid + Sobel gradient, csrc/losses.hip). Mirrors maggie/network/loss.py:7-16 (loss_dtSSD)."""
import torch


def loss_dtSSD(pred, gt, mask):
    dadt = pred[:, 1:] - pred[:, :-1]
    dgdt = gt[:, 1:] - gt[:, :-1]
    diff = (dadt - dgdt) ** 2
    diff = diff * mask[:, 1:]
    return torch.sum(diff) / torch.sum(mask[:, 1:] + 1e-6)
