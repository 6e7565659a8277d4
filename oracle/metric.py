"""TEST INFRASTRUCTURE ONLY (imported by tests/ and tests/golden/make_golden.py, never by the product).

CPU restatement of the reference's validation metrics, maggie/utils/metric.py: SAD :68-78, MSE :80-90, MAD :92-97,
Grad :352-417, dtSSD :422-448 (numpy fp32 like the reference; the Grad convolutions in fp64 numpy instead of torch conv2d).
PINNED by tests/golden/metric_pinned.npz: the reference's own classes run on seeded planes (make_golden.py: metric_fixture)."""
import numpy as np


def _mask(gt, trimap, mode):
    if trimap is None:
        return np.ones_like(gt, dtype=np.float32)
    return ((trimap > 0) if mode == 1 else (trimap == 1)).astype(np.float32)


def _r2(x):
    return x.reshape(-1, *x.shape[-2:])


def sad(pred, gt, trimap=None):
    m = _r2(_mask(gt, trimap, 1))
    return float(np.sum(np.abs(_r2(pred) - _r2(gt)) * m, axis=(1, 2)).sum() * 1e-3), m.shape[0]


def mse(pred, gt, trimap=None):
    m = _r2(_mask(gt, trimap, 1))
    d = ((_r2(pred) - _r2(gt)) ** 2) * m
    return float((np.mean(d, axis=(1, 2)) / (m.sum(axis=(1, 2)) + 1e-6)).sum() * 1e10), m.shape[0]


def mad(pred, gt, trimap=None):
    m = _r2(_mask(gt, trimap, 1))
    d = np.abs(_r2(pred) - _r2(gt)) * m
    return float((np.mean(d, axis=(1, 2)) / (m.sum(axis=(1, 2)) + 1e-6)).sum() * 1e10), m.shape[0]


def gauss_filter(sigma=1.4, epsilon=1e-2):
    half = int(np.ceil(sigma * np.sqrt(-2 * np.log(np.sqrt(2 * np.pi) * sigma * epsilon))))
    size = 2 * half + 1
    g = lambda x: np.exp(-x ** 2 / (2 * sigma ** 2)) / (sigma * np.sqrt(2 * np.pi))      # noqa: E731
    f = np.zeros((size, size))
    for i in range(size):
        for j in range(size):
            f[i, j] = g(i - half) * (-(j - half) * g(j - half) / sigma ** 2)
    return f / np.sqrt((f ** 2).sum())


def _corr2(img, f):
    """F.conv2d(img, f, padding=k//2) for (P, H, W) planes (cross-correlation, zero padding)."""
    k = f.shape[0]
    h = k // 2
    P, H, W = img.shape
    pad = np.zeros((P, H + 2 * h, W + 2 * h), np.float64)
    pad[:, h:h + H, h:h + W] = img
    out = np.zeros((P, H, W), np.float64)
    for i in range(k):
        for j in range(k):
            out += pad[:, i:i + H, j:j + W] * f[i, j]
    return out


def grad(pred, gt, trimap=None):
    m = _r2(_mask(gt, trimap, 1)).astype(np.float64)
    p, g = _r2(pred).astype(np.float32), _r2(gt).astype(np.float32)
    fx = gauss_filter().astype(np.float32).astype(np.float64)
    gn = (g - g.min()) / (g.max() - g.min() + np.float32(1e-6))
    pn = (p - p.min()) / (p.max() - p.min() + np.float32(1e-6))
    mag = lambda x: np.sqrt(_corr2(x, fx) ** 2 + _corr2(x, fx.T) ** 2)                    # noqa: E731
    return float((((mag(gn) - mag(pn)) ** 2) * m).sum() * 0.001), m.shape[0]


def dtssd(pred, gt, trimap=None):
    m = _mask(gt, trimap, 2)
    if pred.ndim == 4:
        pred, gt, m = pred[None], gt[None], m[None]
    e = ((pred[:, 1:] - pred[:, :-1]) - (gt[:, 1:] - gt[:, :-1])) ** 2 * m[:, :-1]
    err = np.sqrt(np.sum(e, axis=(0, 1, 3, 4)))
    return float(np.sum(err) * 0.1), m.shape[2]
