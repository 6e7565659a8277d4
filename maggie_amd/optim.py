"""Optimizer step of the train step on the MI355X: AdamW over one flat parameter buffer, gradient-norm clip folded in.

`FlatAdamW(params, lr, betas, eps, weight_decay, max_grad_norm)` is torch.optim.AdamW (what maggie/engine/optim.py:110-118 builds) for the
case the hot path has -- one parameter group, fp32 parameters on one GPU -- restructured for HBM:

  * on construction every trainable parameter is re-homed into ONE flat fp32 buffer (`p.data` becomes a view; 256-byte aligned
    slots), and so are the two moment buffers;
  * `step()` gathers the gradients into a flat buffer of the same layout (a multi-tensor copy; the hipGraph trunk already hands
    its ~150 gradients over as one flat tensor), then ONE pass `mg_adamw_flat` reduces the gradient norm, scales by
    min(1, max_norm / (norm + 1e-6)) (`clip_grad_norm_` of maggie/engine/train.py:274) and applies the AdamW update:
    28 bytes per parameter of HBM traffic in total, instead of 9 fused multi-tensor launches + 3 norm + 3 scale launches.

`sync_group=True` (or a process group) makes `step()` average the flat gradient buffer over the ranks first -- the whole data-parallel exchange
of the train step as one RCCL all-reduce, instead of DDP's hooks and buckets (call neither DDP nor `parallel.GradSync` then).

`state_dict()` / `load_state_dict()` use torch.optim.AdamW's format (per-parameter `step`, `exp_avg`, `exp_avg_sq`), so the
reference's `last_opt.pth` resumes here and vice versa; torch LR schedulers (OneCycleLR) drive `param_groups[0]['lr']` as usual."""
import math

import torch

from . import hip
from .hip import c_long, c_float, c_int

_ALIGN = 64                                                       # floats: 256-byte slots


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None, sync_group=None):
        # Like torch.optim.AdamW(model.parameters()) (maggie/engine/optim.py:117): EVERY parameter handed in stays in
        # param_groups[0]['params'] -- frozen ones too (SpectralNorm's weight_u / weight_v, dummy_downscale) -- so that the index
        # space of state_dict() is the reference's and its last_opt.pth resumes here (and vice versa). Only the trainable
        # ones live in the flat buffers and are ever updated; torch never creates state for a parameter without a gradient either.
        params = list(params)
        if not any(p.requires_grad for p in params):
            raise ValueError('no trainable parameters')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError('FlatAdamW handles one parameter group')
        params = self._active()
        self.max_grad_norm = max_grad_norm
        self.sync_group = sync_group                              # see step(): data-parallel gradient averaging on the flat buffer
        self.overlap = None                                       # parallel.OverlappedGradSync: chunks already reduced during backward
        self._t = 0
        self._steps = [0] * len(params)
        self.last_grad_norm = None
        self._build()

    # ------------------------------------------------------------------------------------------------------------------
    def _active(self):
        return [p for p in self.param_groups[0]['params'] if p.requires_grad]

    def _build(self):
        ps = self._active()
        if len(ps) != len(self._steps):                           # requires_grad flags changed since construction
            self._steps = [self._t] * len(ps)
        dev = ps[0].device
        hip.need_cuda(ps[0])
        for p in ps:
            if p.dtype != torch.float32 or p.device != dev:
                raise TypeError('FlatAdamW expects fp32 parameters on one device')
        self._offsets, total = [], 0
        for p in ps:
            self._offsets.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self._n = total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self._scratch = torch.zeros(1, dtype=torch.float64, device=dev)
        self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
        view = lambda flat, o, p: flat[o:o + p.numel()].view(p.shape)      # noqa: E731
        self._g_views = [view(self.flat_g, o, p) for o, p in zip(self._offsets, ps)]
        with torch.no_grad():
            for i, (o, p) in enumerate(zip(self._offsets, ps)):
                v = view(self.flat_p, o, p)
                v.copy_(p.data)
                p.data = v                                        # the parameter now lives in the flat buffer
                st = self.state[p]
                old_m, old_v = st.get('exp_avg'), st.get('exp_avg_sq')
                st['exp_avg'], st['exp_avg_sq'] = view(self.flat_m, o, p), view(self.flat_v, o, p)
                if old_m is not None:
                    st['exp_avg'].copy_(old_m)
                    st['exp_avg_sq'].copy_(old_v)
                st['step'] = torch.tensor(float(self._steps[i]))
        self._ptrs = [p.data_ptr() for p in ps]

    def grad_views(self, params):
        """Gradient sink for the captured backward graphs (graphs.GraphedCallable(grad_sink=...), `model.grad_sink = opt.grad_views`): fresh
        views of the flat gradient buffer for `params`, or None when one of them is not managed here. The graphs then write the gradients
        straight into this buffer and step() finds `.grad` already in place."""
        if not self._intact():
            self._build()
        index = {id(p): i for i, p in enumerate(self._active())}
        out = []
        for p in params:
            i = index.get(id(p))
            if i is None:
                return None
            o = self._offsets[i]
            out.append(self.flat_g[o:o + p.numel()].view(p.shape))
        return out

    def grad_runs(self, params):
        """The contiguous stretches of the flat gradient buffer that hold `params` (consecutive parameters of the optimizer's order share a
        stretch, alignment padding included: it stays zero) -- what an in-place all-reduce of a graph's gradients operates on."""
        index = {id(p): i for i, p in enumerate(self._active())}
        idx = sorted(index[id(p)] for p in params if id(p) in index)
        ends = list(self._offsets[1:]) + [self._n]
        runs, k = [], 0
        while k < len(idx):
            j = k
            while j + 1 < len(idx) and idx[j + 1] == idx[j] + 1:
                j += 1
            runs.append(self.flat_g[self._offsets[idx[k]]:ends[idx[j]]])
            k = j + 1
        return runs

    def _intact(self):
        ps = self._active()
        return len(ps) == len(self._ptrs) and all(p.data_ptr() == a for p, a in zip(ps, self._ptrs))

    # ------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._intact():                                    # model.to(...) / .float() moved the parameters: re-home them
            self._build()
        g = self.param_groups[0]
        ps = self._active()
        grads, views, have, absent = [], [], [], []
        for p, gv in zip(ps, self._g_views):
            have.append(p.grad is not None)
            if p.grad is None:
                absent.append(gv)
            elif p.grad.data_ptr() != gv.data_ptr() or p.grad.dtype != torch.float32:
                grads.append(p.grad)                              # (a gradient written into its slot by a graph's gradient sink needs no copy)
                views.append(gv)
        sync = self.sync_group is not None
        # gradient chunks that were all-reduced on the side stream during backward: the main stream must not read them earlier
        done = self.overlap.wait() if (sync and self.overlap is not None) else set()
        if not any(have) and not sync:
            return loss
        if absent:
            torch._foreach_zero_(absent)                          # absent gradients must not count in the norm
        if grads:
            # (contiguous sources keep torch on its one-launch multi-tensor route; a single strided gradient would turn the call into one copy per tensor)
            torch._foreach_copy_(views, [g_ if g_.is_contiguous() else g_.contiguous() for g_ in grads])
        if sync:
            from .parallel import all_reduce_mean
            grp = None if self.sync_group is True else self.sync_group
            if not done:
                # data parallel: ONE all-reduce (mean) of the whole flat gradient buffer over RCCL -- same size and layout on every
                # rank whatever each rank's autograd produced; a parameter without a local gradient contributes zeros, as under DDP
                all_reduce_mean(self.flat_g, grp)
            else:
                # the captured backward graphs already exchanged their chunks on the side stream while backward was running
                # (parallel.OverlappedGradSync); what is left -- parameters outside the graphs this step -- goes in contiguous runs of the
                # flat buffer (same runs on every rank: the ranks see the same geometries in the same order)
                from .parallel import reduce_remaining_runs
                reduce_remaining_runs(self.flat_g, ps, self._offsets, self._n, done, grp)
            have = [True] * len(ps)
        b1, b2 = g['betas']
        clip = self.max_grad_norm is not None
        args = (c_float(g['lr']), c_float(b1), c_float(b2), c_float(g['eps']), c_float(g['weight_decay']))
        tail = (hip.ptr(self._scratch), c_float(float(self.max_grad_norm) if clip else 0.0), hip.ptr(self._norm))

        def launch(lo, hi, t, phases):
            off = 4 * lo
            hip.call('mg_adamw_flat', hip.ctypes.c_void_p(self.flat_p.data_ptr() + off), hip.ctypes.c_void_p(self.flat_g.data_ptr() + off),
                     hip.ctypes.c_void_p(self.flat_m.data_ptr() + off), hip.ctypes.c_void_p(self.flat_v.data_ptr() + off), c_long(hi - lo), *args,
                     c_float(1.0 - b1 ** t), c_float(math.sqrt(1.0 - b2 ** t)), *tail, c_int(phases), hip.stream())

        steps = self._steps
        if all(have) and min(steps) == max(steps):                # the usual case: one launch pair over everything
            t = steps[0] + 1
            launch(0, self._n, t, 3)
            self._steps = [t] * len(ps)
        else:
            # torch.optim.AdamW skips a parameter without a gradient (no decay, no moment update, its own step count): update each
            # contiguous run of parameters that do have one and share a step count; the norm is taken over the whole buffer first
            launch(0, self._n, 1, 1)
            ends = self._offsets[1:] + [self._n]
            i = 0
            while i < len(ps):
                if not have[i]:
                    i += 1
                    continue
                j = i
                while j + 1 < len(ps) and have[j + 1] and steps[j + 1] == steps[i]:
                    j += 1
                launch(self._offsets[i], ends[j], steps[i] + 1, 2)
                for k in range(i, j + 1):
                    steps[k] += 1
                i = j + 1
        self._t = max(self._steps)
        self.last_grad_norm = self._norm                          # device scalar (no sync): total gradient norm before clipping
        return loss

    # ------------------------------------------------------------------------------------------------------------------
    def state_dict(self):
        for p, t in zip(self._active(), self._steps):
            self.state[p]['step'] = torch.tensor(float(t))
        sd = super().state_dict()
        # torch.optim.AdamW holds no state for a parameter it never stepped (no gradient yet): same here
        sd['state'] = {k: v for k, v in sd['state'].items() if float(v.get('step', 0)) > 0}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)                       # replaces the moment tensors by loaded copies ...
        ps = self._active()
        self._steps = [int(float(self.state[p]['step'])) if 'step' in self.state[p] else 0 for p in ps]
        self._t = max(self._steps) if self._steps else 0
        self._build()                                             # ... which _build copies back into the flat buffers
