"""Validation metrics as device reductions -- the classes of maggie/utils/metric.py with the same interface
(`update(pred, gt, trimap=None) -> score / count`, `average()`, `reset()`, `gather_metric()`, `build_metric(names)`), taking the
model's output tensors where they are instead of numpy copies:

    SAD (:68-78), MSE (:80-90), MAD (:92-97), Grad (:352-417), dtSSD (:422-448)

`pred`, `gt`, `trimap`: fp32 device tensors (N, *, H, W) ((T, N_i, H, W) or (B, T, N_i, H, W) for dtSSD). A call costs one pass over
the planes (HIP kernels mg_metric_plane_sums / mg_metric_grad / mg_metric_dtssd) and one small device->host read of fp64 sums; the
reference copies every plane to the host, reduces in numpy and, for Grad, ships the planes back to the GPU.
Conn and MESSDdt (connected components / optical-flow warping on the CPU: skimage, cv2) stay with the caller, as SURVEY 8f says."""
import numpy as np
import torch

from .. import hip
from ..hip import c_int, c_long


def _planes(x):
    hip.need_cuda(x)
    return x.reshape(-1, *x.shape[-2:]).float().contiguous()


class Metric(object):
    mask_mode = 1                                                 # update(): mask = (trimap > 0), metric.py:47

    def __init__(self):
        self.reset()

    def reset(self):
        self.score = 0
        self.count = 0

    def compute_metric(self, pred, gt, trimap, mode, **kargs):
        raise NotImplementedError

    def gather_metric(self, rank=0):
        """Sum of (score, count) over the process group (metric.py:34-41 gathers to `rank`; every rank gets the totals here)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            t = torch.tensor([float(self.score), float(self.count)], dtype=torch.float64)
            if dist.get_backend() == 'nccl':
                t = t.cuda()
            dist.all_reduce(t)
            self.score, self.count = float(t[0]), float(t[1])

    def update(self, pred, gt, trimap=None, **kargs):
        pred, gt = _planes(pred), _planes(gt)
        tri = None if trimap is None else _planes(trimap)
        score, count = self.compute_metric(pred, gt, tri, 0 if tri is None else self.mask_mode, **kargs)
        self.count += count
        self.score += score
        return score * 1.0 / count

    def average(self):
        return self.score / (self.count + 1e-6)


def _plane_sums(pred, gt, tri, mode):
    P, H, W = pred.shape
    out = torch.empty((P, 3), dtype=torch.float64, device=pred.device)
    hip.call('mg_metric_plane_sums', hip.ptr(pred), hip.ptr(gt), hip.ptr(tri), c_int(mode), c_int(P), c_long(H * W), hip.ptr(out), hip.stream())
    return out


class SAD(Metric):
    def compute_metric(self, pred, gt, tri, mode, **kargs):
        s = _plane_sums(pred, gt, tri, mode)
        return float(s[:, 0].sum()) * 1e-3, pred.shape[0]


class MSE(Metric):
    def compute_metric(self, pred, gt, tri, mode, **kargs):
        s = _plane_sums(pred, gt, tri, mode)
        hw = pred.shape[1] * pred.shape[2]
        return float((s[:, 1] / hw / (s[:, 2] + 1e-6)).sum()) * 1e10, pred.shape[0]


class MAD(Metric):
    def compute_metric(self, pred, gt, tri, mode, **kargs):
        s = _plane_sums(pred, gt, tri, mode)
        hw = pred.shape[1] * pred.shape[2]
        return float((s[:, 0] / hw / (s[:, 2] + 1e-6)).sum()) * 1e10, pred.shape[0]


class Grad(Metric):
    def __init__(self):
        super().__init__()
        self.filter_x = self.gauss_filter(1.4).astype(np.float32)

    @staticmethod
    def gauss_filter(sigma, epsilon=1e-2):
        """Gaussian x derivative-of-Gaussian, unit L2 norm (metric.py:365-385); filter_y is its transpose."""
        half = int(np.ceil(sigma * np.sqrt(-2 * np.log(np.sqrt(2 * np.pi) * sigma * epsilon))))
        x = np.arange(-half, half + 1, dtype=np.float64)
        g = np.exp(-x ** 2 / (2 * sigma ** 2)) / (sigma * np.sqrt(2 * np.pi))
        f = np.outer(g, -x * g / sigma ** 2)                      # rows: gaussian(i), columns: dgaussian(j)
        return f / np.sqrt((f ** 2).sum())

    def compute_metric(self, pred, gt, tri, mode, **kargs):
        P, H, W = pred.shape
        if self.filter_x.shape != (9, 9):
            raise hip.MaggieHipError('mg_metric_grad is built for the 9x9 filter of sigma 1.4')
        out = torch.empty(P, dtype=torch.float64, device=pred.device)
        scratch = torch.empty(4, dtype=torch.int32, device=pred.device)
        filt = (hip.ctypes.c_float * 81)(*self.filter_x.reshape(-1).tolist())
        hip.call('mg_metric_grad', hip.ptr(pred), hip.ptr(gt), hip.ptr(tri), c_int(mode), c_int(P), c_int(H), c_int(W), filt, hip.ptr(scratch),
                 hip.ptr(out), hip.stream())
        return float(out.sum()) * 0.001, P


class dtSSD(Metric):
    mask_mode = 2                                                 # mask = (trimap == 1), metric.py:427

    def update(self, pred, gt, trimap=None, **kargs):
        hip.need_cuda(pred)
        if pred.dim() == 4:
            pred, gt = pred[None], gt[None]
            trimap = None if trimap is None else trimap[None]
        B, T, N, H, W = pred.shape
        pred, gt = pred.float().contiguous(), gt.float().contiguous()
        tri = None if trimap is None else trimap.float().contiguous()
        out = torch.empty(N, dtype=torch.float64, device=pred.device)
        hip.call('mg_metric_dtssd', hip.ptr(pred), hip.ptr(gt), hip.ptr(tri), c_int(0 if tri is None else self.mask_mode), c_int(B), c_int(T),
                 c_int(N), c_long(H * W), hip.ptr(out), hip.stream())
        err = float(out.sqrt().sum()) * 0.1
        self.score += err
        self.count += N
        return err / (N + 1e-10)


_DEVICE_METRICS = {'SAD': SAD, 'MSE': MSE, 'MAD': MAD, 'Grad': Grad, 'dtSSD': dtSSD}


def build_metric(metrics):
    """{name: metric object} like metric.py:534-538, for the metrics that live on the device."""
    missing = [m for m in metrics if m not in _DEVICE_METRICS]
    if missing:
        raise NotImplementedError('host-side metrics (skimage / cv2) are not part of this build: %s' % missing)
    return {m: _DEVICE_METRICS[m]() for m in metrics}
