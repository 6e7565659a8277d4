"""Matting losses -- mirror maggie/network/loss.py (loss_dtSSD :7-16,41-44; GradientLoss :67-118; LapLoss :120-191 with its
channels=3-on-1-channel-input quirk: the 5x5 Gaussian is applied as a 1->3 channel conv, so every pyramid level is counted
three times in the numerator while the weight sum in the denominator is counted once)."""
import torch
import torch.nn as nn
from torch.nn import functional as F


def loss_dtSSD(pred, gt, mask):
    dadt = pred[:, 1:] - pred[:, :-1]
    dgdt = gt[:, 1:] - gt[:, :-1]
    diff = (dadt - dgdt) ** 2
    diff = diff * mask[:, 1:]
    return torch.sum(diff) / torch.sum(mask[:, 1:] + 1e-6)


class GradientLoss(nn.Module):
    def __init__(self, eps=1e-6):
        super().__init__()
        kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]]) / 8.0
        self.register_buffer('kernel_x', kx[None, None], persistent=False)
        self.register_buffer('kernel_y', kx.t().contiguous()[None, None], persistent=False)
        self.eps = eps

    def sobel(self, inp):
        n, c, h, w = inp.shape
        p = F.pad(inp.reshape(n * c, 1, h, w), pad=[1, 1, 1, 1], mode='replicate')
        gx = F.conv2d(p, self.kernel_x.to(inp.device), padding=0)
        gy = F.conv2d(p, self.kernel_y.to(inp.device), padding=0)
        return torch.sqrt(gx * gx + gy * gy + self.eps).reshape(n, c, h, w)

    def forward(self, logit, label, mask=None):
        if mask is not None:
            logit = logit * mask
            label = label * mask
            return torch.sum(F.l1_loss(self.sobel(logit), self.sobel(label), reduction='none')) / (mask.sum() + self.eps)
        return F.l1_loss(self.sobel(logit), self.sobel(label), reduction='mean')


def _gauss5():
    k = torch.tensor([[1., 4., 6., 4., 1.], [4., 16., 24., 16., 4.], [6., 24., 36., 24., 6.], [4., 16., 24., 16., 4.],
                      [1., 4., 6., 4., 1.]]) / 256.
    return k


class LapLoss(nn.Module):
    """3-level Laplacian pyramid L1. Evaluated on single-channel maps; the reference's 3x channel replication is folded
    into a factor 3 on each level's numerator (identical value, one third of the memory traffic)."""

    def __init__(self, max_levels=3, channels=3):
        super().__init__()
        self.max_levels = max_levels
        self.channels = channels
        self.register_buffer('gauss', _gauss5()[None, None], persistent=False)

    def _blur(self, img, scale=1.0):
        img = F.pad(img, (2, 2, 2, 2), mode='reflect')
        return F.conv2d(img, self.gauss.to(img.device) * scale)

    def _pyramid(self, img):
        cur = img
        pyr = []
        for _ in range(self.max_levels):
            down = self._blur(cur)[:, :, ::2, ::2]
            up = torch.zeros_like(cur)
            up[:, :, ::2, ::2] = down
            pyr.append(cur - self._blur(up, 4.0))
            cur = down
        return pyr

    def forward(self, input, target, weight=None):
        pi, pt = self._pyramid(input), self._pyramid(target)
        total = 0
        w = weight
        for i in range(self.max_levels):
            d = (pi[i] - pt[i]).abs()
            if w is None:
                total = total + d.mean()
            else:
                total = total + self.channels * (d * w).sum() / (w.sum() + 1e-6)
                w = w[:, :, ::2, ::2]
        return total
