"""Autograd-visible operators of the MaGGIe hot path, built on the HIP kernels (maggie_amd.kernels -> C ABI).

Dense activations are NHWC tensors (N, H, W, C) in the compute dtype (bf16 under torch.autocast, else fp32) with C
padded to a multiple of 8; sparse features are (R, C) row matrices over the sorted active sites. Every forward
AND backward below launches hand-written HIP kernels; PyTorch only provides the autograd graph, memory and streams.
"""
import os

import torch
import torch.distributed as dist

from . import kernels as K
from .kernels import MODE_CONV, MODE_TCONV, MODE_GATHER, ACT_NONE, ACT_RELU, ACT_LRELU  # noqa: F401

LRELU_SLOPE = 0.2
import os as _os
# conv layers with at least this many output rows leave the BatchNorm statistics to the column-statistics kernel (4.5 TB/s, 15 us at 1 M x 32)
# instead of the conv epilogue: there the per-element statistics + butterfly of 8192 small tiles cost more than the extra read (step 13.89 -> 13.72 ms)
UNFUSED_STATS_ROWS = int(_os.environ.get('MAGGIE_UNFUSED_STATS_ROWS', str(1 << 19)))
EXACT_STATS_ROWS = int(_os.environ.get('MAGGIE_EXACT_STATS_ROWS', '32768'))      # BatchNorm layers with at most this many rows use the exact two-pass variance


class ZeroArena:
    """Per-step pool of zero-initialised fp32 scratch (BatchNorm statistic / gradient-sum accumulators). One memset per
    forward replaces ~300 tiny fill kernels; slices are handed out by pointer bump and never outlive the step.

    While a hipGraph is being captured (maggie_amd/graphs.py) slices come from a capture-owned buffer instead: it is
    allocated and zeroed INSIDE the capture (one memset node per graph), so every replay starts from zeros and no captured
    kernel points into the eager arena, which may be re-allocated later."""

    def __init__(self):
        self.buf = None
        self.off = 0
        self.used = 0
        self.need = 1 << 20
        self.cap_buf = None
        self.cap_off = 0
        self.spills = 0                                           # takes a capture's arena could not serve
        self.tally = 0                                            # words handed out so far, eager or captured (graphs.py sizes a graph's arena by it)

    epoch = 0                                                     # bumped once per model forward: stamps per-step prepared weights

    def reset(self, device):
        self.epoch += 1
        del PARKED[:]                                             # leftovers of a backward pass that was abandoned halfway
        for g_ in _OPEN_GROUPS:
            g_.reset()
        del _OPEN_GROUPS[:]
        self.need = max(self.need, self.used + (1 << 16))
        if self.buf is None or self.buf.device != device or self.buf.numel() < self.need:
            self.buf = torch.zeros(self.need, dtype=torch.float32, device=device)
        else:
            self.buf.zero_()
        self.off = 0
        self.used = 0

    def begin_capture(self, device, words=None):
        """Called by the graph builder right after capture starts (forward graph and backward graph each get their own). `words`: what the same
        function took in the warm-up run (`tally` difference): the graph's memset then covers what THIS graph uses, not the largest arena any
        stage ever needed (four fills of 26 MB per step at the headline geometry)."""
        # (+25 % + 1 MB: a few call sites take accumulators only while capturing)
        n = max(self.need, 1 << 20) if words is None else int(words) + int(words) // 4 + (1 << 18)
        self.cap_buf = torch.zeros(n, dtype=torch.float32, device=device)
        self.cap_off = 0
        # the library skips its own fill launches for accumulators inside this (zeroed once per replay) buffer: csrc/common.h mg_zero_words
        K.hip.lib().mg_set_zeroed_range(K.ctypes.c_void_p(self.cap_buf.data_ptr()), K.ctypes.c_long(4 * self.cap_buf.numel()))

    def end_capture(self):
        self.cap_buf = None
        K.hip.lib().mg_set_zeroed_range(None, K.ctypes.c_long(0))

    def acc(self, n, device, dtype=torch.float32):
        """An accumulator of n 4-byte words the callee will clear itself: inside a capture a slice of the graph's zero arena (the callee's fill
        launch is then skipped, mg_set_zeroed_range), else fresh uninitialised memory."""
        if self.cap_buf is not None and torch.cuda.is_current_stream_capturing():
            v = self.take(n, device)
            return v if dtype == torch.float32 else v.view(dtype)
        self.tally += (n + 63) // 64 * 64
        return torch.empty(n, dtype=dtype, device=device)

    def zeros(self, n, device, dtype=torch.float32):
        """Zero-initialised scratch of n 4-byte words: an arena slice inside a capture, torch.zeros otherwise."""
        if self.cap_buf is not None and torch.cuda.is_current_stream_capturing():
            v = self.take(n, device)
            return v if dtype == torch.float32 else v.view(dtype)
        self.tally += (n + 63) // 64 * 64
        return torch.zeros(n, dtype=dtype, device=device)

    def take(self, n, device):
        n_al = (n + 63) // 64 * 64
        self.tally += n_al
        if self.cap_buf is not None and torch.cuda.is_current_stream_capturing():
            if self.cap_off + n_al > self.cap_buf.numel():
                self.spills += 1                                  # (correct, but a fill launch of its own: tests/test_gpu_graphs.py expects none)
                if self.spills == 1:
                    import logging
                    logging.warning('maggie_amd: a captured graph asked its zero arena for more accumulator words than the eager warm-up tallied; '
                                    'the excess is served by fill launches inside the graph (correct, slower).')
                return torch.zeros(n, dtype=torch.float32, device=device)
            v = self.cap_buf[self.cap_off:self.cap_off + n]
            self.cap_off += n_al
            return v
        self.used += n_al
        if self.buf is None or self.buf.device != device or self.off + n_al > self.buf.numel():
            return torch.zeros(n, dtype=torch.float32, device=device)
        v = self.buf[self.off:self.off + n]
        self.off += n_al
        return v


def flush_bn_count_log(log):
    """num_batches_tracked += (number of calls) for every logged BatchNorm: one foreach launch per distinct multiplicity (the temporal decoder
    calls some layers two or more times per forward; 81 one-element add kernels per video step otherwise)."""
    counts, tensors = {}, {}
    for t in log:
        counts[id(t)] = counts.get(id(t), 0) + 1
        tensors[id(t)] = t
    by_c = {}
    for k, c in counts.items():
        by_c.setdefault(c, []).append(tensors[k])
    for c, ts in by_c.items():
        torch._foreach_add_(ts, c)


ARENA = ZeroArena()
K.ACC = ARENA.acc                 # kernels.py allocates the accumulators its callees clear through the arena (no fill launch inside a capture)
CAPTURE_FIXUPS = []            # (slice of CAPTURE_TABLE, host tensor to upload once the capture has ended)
CAPTURE_TABLE = [None, 0]      # [int64 device buffer owned by the graph being captured, bump offset]


def capture_table(n):
    buf, off = CAPTURE_TABLE
    if buf is None or off + n > buf.numel():
        raise RuntimeError('capture table exhausted')
    CAPTURE_TABLE[1] = off + n
    return buf[off:off + n]
EAGER_TOKEN_CHECK = True       # InstanceMatteDecoder checks its tokens for NaN itself (a host sync) unless MaGGIe.forward, which reads all step flags at once, clears this
DEFER_BN_COUNTERS = False      # set by MaGGIe.forward: num_batches_tracked of all BN layers is bumped by one foreach op per step
BN_COUNT_LOG = None            # a list while a stage of the video model runs: BatchNorm calls are logged, the stage bumps the counters with one foreach op per multiplicity


FP16_AUTOCAST_AS_BF16 = os.environ.get('MAGGIE_FP16_AUTOCAST', '') == 'bf16'
_FP16_WARNED = False


def compute_dtype():
    """The kernels' 16-bit storage type inside torch.autocast(device_type='cuda'): bf16 (this build's choice, BASELINE north_star) or fp16 (the
    reference's `--precision 16`: torch.cuda.amp fp16 autocast + GradScaler, engine/train.py:208,227-229 -- served by the fp16 instantiations
    of the same kernels, v_mfma_f32_16x16x32_f16, fp32 accumulate); fp32 otherwise. MAGGIE_FP16_AUTOCAST=bf16 serves fp16 autocast with the
    bf16 kernels instead (fp32's exponent range: the loss scaler keeps working but is not needed; 8 mantissa bits instead of 11)."""
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype('cuda') if hasattr(torch, 'get_autocast_dtype') else torch.get_autocast_gpu_dtype()
        if dt == torch.float16:
            if not FP16_AUTOCAST_AS_BF16:
                return torch.float16
            global _FP16_WARNED
            if not _FP16_WARNED:
                _FP16_WARNED = True
                import warnings
                warnings.warn('MaGGIe (MI355X build): fp16 autocast is served by the bf16 kernels (MAGGIE_FP16_AUTOCAST=bf16); the GradScaler '
                              'keeps working but is not needed.')
            return torch.bfloat16
        if dt != torch.bfloat16:
            raise K.hip.MaggieHipError('MaGGIe (MI355X build): autocast dtype %s is not supported -- the HIP kernels compute in bf16 or fp16 '
                                       '(fp32 accumulate) under autocast, fp32 without.' % dt)
        return torch.bfloat16
    return torch.float32


def pad8(c):
    return (c + 7) // 8 * 8


# ----------------------------------------------------------------------------------------------------------------------
# weight layout plumbing (tiny tensors; plain torch so gradients flow back to the OIHW / KRSC parameters)
# ----------------------------------------------------------------------------------------------------------------------

def weight_oihw_to_krsc(w, dtype, cin_pad=None, cout_pad=None):
    """(Cout, Cin, R, S) -> (Cout_pad, R*S, Cin_pad) in `dtype`."""
    co, ci, r, s = w.shape
    w = w.permute(0, 2, 3, 1).reshape(co, r * s, ci)
    return _pad_krsc(w, dtype, cin_pad, cout_pad)


def weight_iohw_to_krsc(w, dtype, cin_pad=None, cout_pad=None):
    """ConvTranspose2d weight (Cin, Cout, R, S) -> (Cout_pad, R*S, Cin_pad)."""
    ci, co, r, s = w.shape
    w = w.permute(1, 2, 3, 0).reshape(co, r * s, ci)
    return _pad_krsc(w, dtype, cin_pad, cout_pad)


def weight_krsc_param(w, dtype, cin_pad=None, cout_pad=None):
    """spconv-layout parameter (Cout, R, S, Cin) -> (Cout_pad, R*S, Cin_pad)."""
    co, r, s, ci = w.shape
    return _pad_krsc(w.reshape(co, r * s, ci), dtype, cin_pad, cout_pad)


def _pad_krsc(w, dtype, cin_pad, cout_pad):
    co, taps, ci = w.shape
    cin_pad = pad8(ci) if cin_pad is None else cin_pad
    cout_pad = co if cout_pad is None else cout_pad
    if cin_pad != ci or cout_pad != co:
        w = torch.nn.functional.pad(w, (0, cin_pad - ci, 0, 0, 0, cout_pad - co))
    return w.to(dtype).contiguous()


# ----------------------------------------------------------------------------------------------------------------------
# side streams: work that is off the data-gradient critical path (the weight-gradient GEMMs of the trunk) can be issued on a
# second HIP stream, so that inside the captured backward graph it forms a parallel branch: a deep-layer dgrad GEMM (one
# workgroup per CU) and the previous layer's wgrad GEMM share the chip instead of running back to back.
# Measured (tools/micro_overlap.py): 97 -> 66 us for a 16x16x512 layer pair, 125 -> 110 us at 32x32x512, nothing from 64x64 up
# -- and nothing on the whole backward graph (9.2-9.3 ms either way: the deep layers are a small share). Off by default.
# ----------------------------------------------------------------------------------------------------------------------
SIDE_WGRAD = os.environ.get('MAGGIE_SIDE_WGRAD', '0') == '1'
# Parked slab reductions (round 4): a conv whose weight came out of the batched weight pipeline or the weight bank does not launch its
# "row-split slabs -> dW" kernel; the descriptors wait in PARKED and ONE launch runs them all where the weight gradients meet anyway
# (SpectralNormBatch.backward / WeightBank.backward) -- ~80 launches of a few microseconds each per training step become one or two.
# Only weights used by exactly one convolution call of the step are parked (autograd would add the gradients of a second use before the join).
PARK_WGRAD = os.environ.get('MAGGIE_PARK_WGRAD', '1') != '0'
PARKED = []
# id(parameter) -> fp32 tensor its gradient should be WRITTEN to (set by graphs.GraphedCallable around the backward capture when the optimizer
# offers a gradient sink). Producers that know it (SpectralNormBatch.backward) write there and return that very tensor as the gradient.
GRAD_DEST = {}


def flush_parked():
    if PARKED:
        K.wgrad_reduce_batched(PARKED)


def _parked_must_be_flushed():
    """Runs when the autograd engine has finished the backward pass (or the captured backward) in which something was parked: every parked slab
    reduction must have met its join (SpectralNormBatch.backward / WeightBank.backward) INSIDE that pass -- otherwise a dW left this pass as
    unreduced slabs (a graph split that puts the join into another graph would replay the GEMMs without ever reducing them). ADVICE round 4."""
    if _OPEN_GROUPS:
        n = len(_OPEN_GROUPS)
        for g in _OPEN_GROUPS:
            g.reset()
        del _OPEN_GROUPS[:]
        del PARKED[:]
        raise K.hip.MaggieHipError('%d weight(s) used by several convolution calls did not see the backward of all their uses in this pass: their '
                                   'gradients are incomplete. Set MAGGIE_GROUP_WGRAD=0 if some uses are differentiated separately.' % n)
    if PARKED:
        n = len(PARKED)
        del PARKED[:]
        raise K.hip.MaggieHipError('%d parked weight-gradient slab reduction(s) were never flushed: the backward pass that parked them ended before '
                                   'the join of their weights (SpectralNormBatch / WeightBank backward) ran -- the weight gradients of this pass are '
                                   'invalid. Set MAGGIE_PARK_WGRAD=0 if the weight pipeline and its convolutions are differentiated separately.' % n)


def _park_list():
    """-> PARKED, with the end-of-backward check armed on the first parking of a backward pass."""
    if not PARKED:
        torch.autograd.Variable._execution_engine.queue_callback(_parked_must_be_flushed)
    return PARKED


class _UseGroup:
    """A joined weight consumed by SEVERAL convolution calls of one step (the ConvGRU gate weights: one call per frame and direction). When all
    the calls are plain convolutions of ONE geometry, their weight-gradient GEMMs write their row-split slabs side by side into one workspace
    and the LAST of them to run parks ONE reduction over all the slabs -- instead of one reduce launch per call plus the autograd engine's
    pairwise adds of the per-call dW (video step, T = 3, bidirectional: 10 reduce launches of 14 us + 8 adds; and the sum is formed in fp32 once
    instead of being rounded to the weight's 16-bit type per call). The calls before the last return no gradient of their own."""
    __slots__ = ('geoms', 'done', 'ws', 'need', 'out', 'desc', 'plain')

    def __init__(self):
        self.geoms = []
        self.reset()

    def reset(self):
        self.done, self.ws, self.need, self.out, self.desc, self.plain = 0, None, 0, None, None, False

    def uniform(self):
        g = self.geoms
        return len(g) > 1 and g[0] is not None and all(q == g[0] for q in g)


GROUP_WGRAD = os.environ.get('MAGGIE_GROUP_WGRAD', '1') != '0'     # 0: every call reduces its own slabs, autograd adds the dW (A/B switch)
_OPEN_GROUPS = []


def _group_wgrad(grp, call):
    """`call(out, park, park_ws)` runs this use's K.conv_wgrad. -> this use's contribution to dW: None for every use but the one that completes the
    group (or every use's own dW when the geometry turns out to need no slab reduction)."""
    k = len(grp.geoms)
    if grp.plain:
        grp.done += 1
        if grp.done == k:
            grp.reset()
        return call(None, None, None)
    first = grp.done == 0
    tmp = []

    def park_ws(need, device):
        if grp.ws is None:
            grp.need, grp.ws = need, torch.empty(k * need, dtype=torch.float32, device=device)
        if need != grp.need:
            raise K.hip.MaggieHipError('grouped weight gradient: the workspace size changed between the uses of one weight (%d != %d)' % (need, grp.need))
        return grp.ws[grp.done * need:(grp.done + 1) * need]

    out = call(grp.out, tmp, park_ws)
    if first:
        d = tmp[0][0] if tmp else None
        if d is None or d.splits * d.n != grp.need or d.ws != grp.ws.data_ptr():
            # no slab reduction for this geometry (dW was written directly), or slabs that do not tile the workspace: every use on its own
            if tmp:
                K.wgrad_reduce_batched(tmp)
            grp.reset()
            grp.plain, grp.done = True, 1
            if k == 1:
                grp.reset()
            return out
        grp.out, grp.desc = out, d
        if not _OPEN_GROUPS and not PARKED:
            torch.autograd.Variable._execution_engine.queue_callback(_parked_must_be_flushed)
        _OPEN_GROUPS.append(grp)
    grp.done += 1
    if grp.done < k:
        return None
    d = K.hip.WgradParked()
    for f, _ in d._fields_:
        setattr(d, f, getattr(grp.desc, f))
    d.splits = grp.desc.splits * k                                # the k calls' slabs lie back to back: one reduction over all of them
    _park_list().append((d, grp.ws, grp.out))
    out = grp.out
    _OPEN_GROUPS.remove(grp)
    grp.reset()
    return out


def _count_use(w, geom=None, needs_grad=True):
    """-> the per-step use counter of a joined weight (a one-element list shared by every autograd Function of this module that consumes `w`:
    ConvRaw in every mode -- also the transposed and side-stream forms, which never park themselves -- and GatherConv), or None. A weight is
    parked only when this counter says its dW has ONE producer: autograd would add a second producer's gradient into the unreduced slabs.
    `geom`: what a plain convolution call looks like to the grouped form above (None: this use can not take part)."""
    if not (PARK_WGRAD and getattr(w, '_mg_join', False)):
        return None
    if not needs_grad:
        #  a use whose backward can never run -- under no_grad, or of a frozen weight -- must not be counted: every counted use but
        # the last returns dW = None and leaves its slabs to the group, which a missing backward would keep open for ever. (`needs_grad` =
        # ctx.needs_input_grad of the weight: inside an autograd Function's forward torch.is_grad_enabled() is always False.)
        return None
    cnt = getattr(w, '_mg_uses', None)
    if cnt is None:
        cnt = [0]
        w._mg_uses = cnt
        w._mg_group = _UseGroup()
    cnt[0] += 1
    w._mg_group.geoms.append(geom)
    return cnt
_SIDE_STREAMS = {}
_FORKED = []


def side_stream(device, idx=0):
    key = (device.index, idx)
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device)
    return s


def fork_side(device, idx=0):
    """-> (side stream ordered after everything issued so far on the current stream, the current stream)."""
    main = torch.cuda.current_stream(device)
    side = side_stream(device, idx)
    side.wait_stream(main)
    if side not in _FORKED:
        _FORKED.append(side)
    return side, main


# The encoder's shortcut branches on a side stream (round 5): conv -> ReLU -> BN twice per branch, needed only by the detail stage, next to the
# backbone -> ASPP -> decoder -> instance-token chain (a long string of 10-token kernels that leaves the chip almost idle) -- forward and, through
# autograd's per-node streams, backward; inside a captured hipGraph a parallel branch with ONE join. The library's deterministic sums stay
# deterministic: the side stream has a slot scratch of its own (csrc/det.hip: mg_det_side_stream). MAGGIE_SIDE_SHORTCUTS=0 keeps one stream.
SIDE_SHORTCUTS = os.environ.get('MAGGIE_SIDE_SHORTCUTS', '0') != '0'
_SIDE_LANE = {}


def side_lane(device):
    """-> the registered side stream of `device` (created and registered with the library on first use, outside any capture), or None."""
    if not (SIDE_SHORTCUTS and device.type == 'cuda'):
        return None
    key = device.index if device.index is not None else torch.cuda.current_device()
    s = _SIDE_LANE.get(key)
    if s is None:
        if torch.cuda.is_current_stream_capturing():
            return None                                           # (first use inside a capture: the scratch cannot be allocated now)
        s = side_stream(device, 3)
        with torch.cuda.device(device):
            K.hip.call('mg_det_side_stream', K.ctypes.c_void_p(s.cuda_stream), K.c_long(64 << 20))
        _SIDE_LANE[key] = s
    return s


class Deferred:
    """A launch sequence to be issued later (`run()`: on the side lane) together with the tensors it reads."""
    __slots__ = ('fn', 'tensors')

    def __init__(self, fn, *tensors):
        self.fn, self.tensors = fn, tensors

    def run(self):
        return on_side_lane(self.fn, *self.tensors)


def on_side_lane(fn, *tensors):
    """fn() on the device's side stream, ordered after everything issued so far on the current stream; the current stream is NOT made to wait
    (join_side() does that where the results are first needed). `tensors`: inputs allocated on the current stream that fn reads."""
    dev = tensors[0].device
    side = side_lane(dev) if torch.is_grad_enabled() else None
    if side is None:
        return fn()
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    if side not in _FORKED:
        _FORKED.append(side)
    for t in tensors:
        t.record_stream(side)
    with torch.cuda.stream(side):
        out = fn()
    for t in (out if isinstance(out, (tuple, list)) else (out,)):
        if torch.is_tensor(t):
            t.record_stream(main)
    return out


def join_side():
    """The current stream waits for every side stream that was forked since the last join."""
    if _FORKED:
        main = torch.cuda.current_stream()
        for s in _FORKED:
            main.wait_stream(s)
        del _FORKED[:]


def parallel_branches(fns, device, tag=''):
    """Run independent launch sequences concurrently: fns[0] on the current stream, the others on side streams forked from it, all joined
    before returning. -> list of results. The chains this is used for (the three scales of the matting losses, the ASPP branches) are
    strings of small kernels at the latency floor of a launch: one after the other their floors add up, side by side they overlap --
    inside a captured hipGraph they become parallel branches. Tensors produced on a side stream are registered with the caching
    allocator for the current stream (they are consumed there after the join). MAGGIE_BRANCHES=0 runs them in sequence."""
    if len(fns) < 2 or not (PAR_BRANCHES == '1' or (tag and tag in PAR_BRANCHES.split(','))):
        return [f() for f in fns]
    main = torch.cuda.current_stream(device)
    sides = []
    for i in range(1, len(fns)):
        sd = side_stream(device, 8 + i)
        sd.wait_stream(main)
        sides.append(sd)
    outs = [None] * len(fns)
    for i, sd in enumerate(sides, 1):
        with torch.cuda.stream(sd):
            outs[i] = fns[i]()
    outs[0] = fns[0]()
    for i, sd in enumerate(sides, 1):
        main.wait_stream(sd)
        r = outs[i]
        for t in (r if isinstance(r, (tuple, list)) else (r,)):
            if torch.is_tensor(t):
                t.record_stream(main)
    return outs


# Off by default: measured on the frozen workload (round 3), the three loss scales + the five ASPP branches as parallel graph branches
# made the step 2.0 ms SLOWER (14.45 -> 16.44 ms) -- fork / join edges of a replayed hipGraph cost far more than the launch floors they hide.
PAR_BRANCHES = os.environ.get('MAGGIE_BRANCHES', '0')        # '0' | '1' | comma list of tags ('loss', 'aspp')
if K.hip.DETERMINISTIC and (SIDE_WGRAD or PAR_BRANCHES != '0'):
    # the ordered sums stage their partials in ONE library-owned scratch (csrc/det.hip): kernels of this library must then run on one stream at a time
    raise K.hip.MaggieHipError('MAGGIE_SIDE_WGRAD / MAGGIE_BRANCHES run library kernels on concurrent streams, which share the slot scratch of the deterministic '
                               'sums: set MAGGIE_DETERMINISTIC=0 with them')


class WeightBankPlan:
    """Static description of a set of small parameters converted together (csrc/weight_bank.hip). `items`: list of
    (parameter, (cout, taps, cin), cout_pad, cin_pad, flip_t, is_bias); a bias is (1, 1, C) and stays fp32."""
    MAX = 64

    def __init__(self, items):
        if len(items) > self.MAX:
            raise K.hip.MaggieHipError('weight bank holds at most %d parameters, got %d' % (self.MAX, len(items)))
        self.items = items
        self.n = len(items)
        self.fwd, self.bwd = (K.hip.WbEntry * self.n)(), (K.hip.WbEntry * self.n)()
        self.w_off, self.b_off, self.g_off = [], [], []
        wo = bo = go = 0
        for i, (p, (co, taps, ci), co_pad, ci_pad, flip_t, is_bias) in enumerate(items):
            for e in (self.fwd[i], self.bwd[i]):
                e.cout, e.taps, e.cin, e.cout_pad, e.cin_pad, e.flip_t = co, taps, ci, co_pad, ci_pad, flip_t
            n_out = co_pad * taps * ci_pad
            self.w_off.append(None if is_bias else wo)
            self.b_off.append(bo if is_bias else None)
            self.g_off.append(go)
            if is_bias:
                bo += (n_out + 3) // 4 * 4
            else:
                wo += (n_out + 7) // 8 * 8
            go += (co * taps * ci + 3) // 4 * 4
        self.total_w, self.total_b, self.total_g = wo, bo, go


class WeightBank(torch.autograd.Function):
    """All parameters of `plan` -> kernel layouts in one launch; all their gradients back in one launch."""

    @staticmethod
    def forward(ctx, plan, dtype, *params):
        dev = params[0].device
        want_t = any(ctx.needs_input_grad[2:])
        code = K.hip.code_of(dtype)
        esz = 4 if dtype == torch.float32 else 2
        w = torch.empty(plan.total_w, dtype=dtype, device=dev)
        wt = torch.empty(plan.total_w, dtype=dtype, device=dev) if want_t else None
        b = torch.empty(max(plan.total_b, 1), dtype=torch.float32, device=dev)
        wp, wtp, bp = w.data_ptr(), (wt.data_ptr() if want_t else 0), b.data_ptr()
        outs = []
        for i, (it, p) in enumerate(zip(plan.items, params)):
            e = plan.fwd[i]
            (co, taps, ci), co_pad, ci_pad, is_bias = it[1], it[2], it[3], it[5]
            e.src = p.data_ptr()
            n_out = co_pad * taps * ci_pad
            if is_bias:
                o = plan.b_off[i]
                e.dst, e.dst_t, e.dtype = bp + 4 * o, None, K.hip.F32
                outs.append(b[o:o + n_out])
            else:
                o = plan.w_off[i]
                e.dst, e.dst_t, e.dtype = wp + esz * o, (wtp + esz * o) if want_t else None, code
                v = w[o:o + n_out].view(co_pad, taps, ci_pad)
                if want_t:
                    v._mg_wt = wt[o:o + n_out].view(ci_pad, taps, co_pad)
                v._mg_join = True                                 # its dW comes back to backward() below
                outs.append(v)
        K.hip.call('mg_weight_bank', plan.fwd, K.c_int(plan.n), K.c_int(0), K.hip.stream())
        ctx.plan, ctx.code, ctx.dtype = plan, code, dtype
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        plan = ctx.plan
        flush_parked()                                            # the dW tensors below may still be slabs
        dev = next(g for g in grads if g is not None).device
        flat = torch.empty(plan.total_g, dtype=torch.float32, device=dev)
        fp = flat.data_ptr()
        keep, outs = [], []
        for i, (it, g) in enumerate(zip(plan.items, grads)):
            e = plan.bwd[i]
            (co, taps, ci), is_bias = it[1], it[5]
            want = torch.float32 if is_bias else ctx.dtype
            if g is not None and (g.dtype != want or not g.is_contiguous()):
                g = g.to(want).contiguous()
            keep.append(g)
            e.src, e.dst, e.dst_t = (None if g is None else g.data_ptr()), fp + 4 * plan.g_off[i], None
            e.dtype = K.hip.F32 if is_bias else ctx.code
            outs.append(flat[plan.g_off[i]:plan.g_off[i] + co * taps * ci].view(it[0].shape) if ctx.needs_input_grad[2 + i] else None)
        K.hip.call('mg_weight_bank', plan.bwd, K.c_int(plan.n), K.c_int(1), K.hip.stream())
        return (None, None) + tuple(outs)


def weight_bank(plan, dtype, params=None):
    """`params`: the LIVE parameter tensors in plan order (default: the ones seen when the plan was built). Callers that may run inside a
    graph capture pass them: the modules then hold stand-in leaves (graphs._ParamAliases) and the captured backward must be
    differentiated with respect to those."""
    return WeightBank.apply(plan, dtype, *(params if params is not None else [it[0] for it in plan.items]))


class SpectralNormWeight(torch.autograd.Function):
    """W_bar / sigma after ONE power iteration (u, v updated in place, no gradient through them), emitted directly in the
    conv kernels' layout and compute dtype -- one fused HIP pipeline instead of torch.mv/norm/dot per wrapped conv."""

    @staticmethod
    def forward(ctx, w_bar, u, v, transposed, dtype, pad_in):
        w = w_bar.detach().contiguous()
        out, work = K.spectral_norm(w, u, v, transposed, dtype, pad_in)
        ctx.save_for_backward(w, u.detach().clone(), v.detach().clone(), work)
        ctx.transposed = transposed
        return out

    @staticmethod
    def backward(ctx, G):
        w, u, v, work = ctx.saved_tensors
        return K.spectral_norm_bwd(G, w, u, v, ctx.transposed, work), None, None, None, None, None


def spectral_norm_weight(w_bar, u, v, transposed, dtype, pad_in):
    return SpectralNormWeight.apply(w_bar, u.data, v.data, transposed, dtype, pad_in)


def _weight_source(m):
    """(W, u, v, transposed, in_channels, plain) of one entry of the batched weight pipeline: a SpectralNorm wrapper, or an ordinary
    conv holder (nn.Conv2d-like: OIHW `weight`, `in_channels`) that rides along for the layout / dtype conversion only."""
    inner = getattr(m, 'module', None)
    if inner is not None and hasattr(inner, 'weight_bar'):
        return inner.weight_bar, inner.weight_u, inner.weight_v, int(inner.transposed), inner.in_channels, 0
    return m.weight, None, None, 0, m.in_channels, 1


class SpectralNormPlan:
    """Descriptor table + work lists for the batched SpectralNorm kernels (built once per model / dtype / device)."""

    def __init__(self, modules, dtype):
        import ctypes
        import numpy as np
        self.modules = modules
        self.dtype = dtype
        self.sources = [_weight_source(m) for m in modules]                # used below only; tensors are dropped at the end
        dev = self.sources[0][0].device
        n = len(modules)
        descs = (K.hip.SnDesc * n)()
        out_off = work_off = dw_off = 0
        k1, k2, k3 = [], [], []
        self.shapes, self.out_slices, self.dw_slices = [], [], []
        for c, (w, u, v, transposed, cin, plain) in enumerate(self.sources):
            A, B, kh, kw = w.shape
            taps = kh * kw
            if taps > 16:
                raise K.hip.MaggieHipError(f'batched SpectralNorm tiles hold up to 16 taps (SN_MAXTAPS), got a {kh}x{kw} kernel')
            pad_in = pad8(cin)
            cout = B if transposed else A
            d = descs[c]
            d.W, d.u, d.v = w.data_ptr(), (None if plain else u.data_ptr()), (None if plain else v.data_ptr())
            d.out_off, d.work_off, d.dw_off = out_off, work_off, dw_off
            d.A, d.B, d.taps, d.transposed, d.pad_in, d.plain = A, B, taps, transposed, pad_in, plain
            Wd = B * taps
            n_out = cout * taps * pad_in
            self.shapes.append((cout, taps, pad_in, tuple(w.shape)))
            self.out_slices.append((out_off, n_out))
            self.dw_slices.append((dw_off, w.numel()))
            if not plain:                                         # power iteration work items
                for cb in range((Wd + 31) // 32):                 # W^T u: a workgroup owns 32 columns over all rows (fixed-order sums)
                    k1.append((c, cb, 0, 0))
                for rg in range((A + 3) // 4):
                    k2.append((c, rg, 0, 0))
            d.k3_first = len(k3)
            for at in range((A + 15) // 16):                      # (SN_TA x SN_TB) parameter tiles, csrc/spectral_norm.hip
                for bt in range((B + 31) // 32):
                    k3.append((c, at, bt, 0))
            d.k3_count = len(k3) - d.k3_first
            out_off += (n_out + 7) // 8 * 8
            work_off += (Wd + A + 4 + 3) // 4 * 4
            dw_off += (w.numel() + 3) // 4 * 4
        self.n, self.total_out, self.total_work, self.total_dw = n, out_off, work_off, dw_off
        raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).clone()
        self.descs = raw.to(dev)
        mk = lambda lst: torch.from_numpy(np.asarray(lst, np.int32).reshape(-1, 4)).to(dev)
        self.k1, self.k2, self.k3 = mk(k1), mk(k2), mk(k3)
        self.ptr_key = self._ptrs()
        self.sources = [(None, None, None) + tuple(src[3:]) for src in self.sources]      # metadata only: never hand out a stale tensor

    def _ptrs(self):
        # LIVE tensors of the modules (not the ones seen when the plan was built): parameters may have been re-allocated or re-homed
        # since, and during a graph capture the modules hold stand-ins
        out = []
        for m in self.modules:
            w, u, v = _weight_source(m)[:3]
            out.append((w.data_ptr(), 0 if u is None else u.data_ptr(), 0 if v is None else v.data_ptr()))
        return tuple(out)

    def valid(self, dtype):
        return dtype == self.dtype and self._ptrs() == self.ptr_key


class SpectralNormBatch(torch.autograd.Function):
    """All spectrally-normalised weights of a model in one batched HIP pipeline (forward: 5 launches; backward: 2)."""

    @staticmethod
    def forward(ctx, plan, *w_bars):
        dev = w_bars[0].device
        out = torch.empty(plan.total_out, dtype=plan.dtype, device=dev)
        want_t = any(ctx.needs_input_grad[1:])                    # a backward pass will need the transposed twins
        out_t = torch.empty(plan.total_out, dtype=plan.dtype, device=dev) if want_t else None
        work = torch.empty(plan.total_work, dtype=torch.float32, device=dev)
        hipc, c_int = K.hip.call, K.c_int
        hipc('mg_spectral_norm_batched', K.hip.ptr(plan.descs), c_int(plan.n), K.hip.ptr(plan.k1), c_int(plan.k1.shape[0]), K.hip.ptr(plan.k2),
             c_int(plan.k2.shape[0]), K.hip.ptr(plan.k3), c_int(plan.k3.shape[0]), K.hip.ptr(work), K.c_long(plan.total_work), K.hip.ptr(out),
             K.hip.ptr(out_t), c_int(K.hip.dtype_code(out)), K.hip.stream())
        ctx.plan = plan
        ctx.w_ids = tuple(id(w) for w in w_bars)                  # GRAD_DEST lookups in backward()
        ctx.save_for_backward(work)
        outs = tuple(out[o:o + n].view(sh[0], sh[1], sh[2]) for (o, n), sh in zip(plan.out_slices, plan.shapes))
        for t_, src in zip(outs, plan.sources):
            t_._mg_side_wgrad = True                              # every dW of these weights meets again in backward() below: the join point
            t_._mg_join = True
            t_._mg_cin = src[4]                                   # real (unpadded) input channels: algorithmic-FLOP accounting of the bench
        if out_t is not None:
            # (Cin_pad, taps, Cout) twins for the data-gradient convolution ride along as a Python attribute of each weight (ConvTranspose
            # weights too since round 5: their twin was a strided permute + copy of up to 8 MB per step in ConvRaw.backward)
            for t_, (o, n), sh, src in zip(outs, plan.out_slices, plan.shapes, plan.sources):
                t_._mg_wt = out_t[o:o + n].view(sh[2], sh[1], sh[0])
        return outs

    @staticmethod
    def backward(ctx, *grads):
        plan = ctx.plan
        (work,) = ctx.saved_tensors
        dev = work.device
        join_side()                                               # the weight-gradient GEMMs ran on the side stream
        flush_parked()                                            # ... and their slab reductions were parked until here
        n = len(grads)
        keep, table = [], [0] * (2 * n)
        for i, (g, src) in enumerate(zip(grads, plan.sources)):
            if g is None:
                keep.append(None)
                continue
            if g.dtype != plan.dtype:
                g = g.to(plan.dtype)
            if src[3] and g.dim() == 3 and not g.is_contiguous() and g.permute(2, 1, 0).is_contiguous():
                # a ConvTranspose weight's gradient as its role-swapped GEMM wrote it, (Cin_pad, taps, Cout): read in that layout (bit 0 of the entry)
                g = g.permute(2, 1, 0)
                table[i] = g.data_ptr() | 1
            else:
                g = g.contiguous()
                table[i] = g.data_ptr()
            keep.append(g)
        # destinations named by the caller (graphs.GraphedCallable with a gradient sink: slices of the optimizer's flat gradient buffer) -- the
        # kernel writes there, the capture's closing multi-tensor copy has nothing left to move for these parameters (~110 MB per step)
        dests = [None] * n
        if GRAD_DEST:
            for i, (wid, (_o, cnt)) in enumerate(zip(ctx.w_ids, plan.dw_slices)):
                d = GRAD_DEST.get(wid)
                if d is not None and d.dtype == torch.float32 and d.is_contiguous() and d.numel() == cnt and d.device == dev:
                    dests[i] = d
                    table[n + i] = d.data_ptr()
        host_ptrs = torch.tensor(table, dtype=torch.int64)
        if torch.cuda.is_current_stream_capturing():
            # no host->device traffic inside a capture: the table is carved from a buffer the graph builder allocated
            # BEFORE the capture (memory allocated during a capture is recycled between the nodes of the graphs sharing
            # its pool, so it cannot hold constants) and is filled once, right after the capture ends -- the gradient
            # addresses are fixed for the life of the graph
            ptrs = capture_table(2 * n)
            CAPTURE_FIXUPS.append((ptrs, host_ptrs))
        else:
            ptrs = host_ptrs.to(dev, non_blocking=True)
        dW = torch.empty(plan.total_dw, dtype=torch.float32, device=dev)
        dot_part = torch.empty(plan.k3.shape[0], dtype=torch.float32, device=dev)      # per-tile partials of <G, W>, added in tile order
        K.hip.call('mg_spectral_norm_batched_bwd_to', K.hip.ptr(plan.descs), K.c_int(plan.n), K.hip.ptr(plan.k3), K.c_int(plan.k3.shape[0]),
                   K.hip.ptr(ptrs), K.c_int(K.hip.code_of(plan.dtype)), K.hip.ptr(work), K.hip.ptr(dW),
                   K.hip.ptr(ptrs[n:]) if any(d is not None for d in dests) else None, K.hip.ptr(dot_part), K.hip.stream())
        outs = tuple((dW[o:o + cnt] if d is None else d).view(sh[3]) for (o, cnt), sh, d in zip(plan.dw_slices, plan.shapes, dests))
        return (None,) + outs


def spectral_norm_prepare(modules, dtype, cache):
    """Batched power iteration + normalisation for `modules` (list of SpectralNorm); stores each result on the module
    (`_prepared`) for its next `krsc()` call. `cache` is a dict owned by the model holding the plan."""
    if not modules:
        return
    plan = cache.get('plan')
    if plan is None or not plan.valid(dtype):
        plan = SpectralNormPlan(modules, dtype)
        cache['plan'] = plan
    # the weight tensors are looked up NOW, not taken from the plan: while a graph is being captured the modules hold stand-in
    # parameters (graphs._ParamAliases) and the captured backward must be differentiated with respect to those
    outs = SpectralNormBatch.apply(plan, *[_weight_source(m)[0] for m in modules])
    for m, o in zip(modules, outs):
        m.__dict__['_prepared'] = o
        m.__dict__['_prepared_epoch'] = ARENA.epoch


def plain_krsc(conv, dtype, keep=False):
    """(Cout, taps, Cin_pad) weight of an ordinary conv holder: the tensor the batched weight pipeline prepared for this step (with its
    dgrad twin attached) when there is one, else converted here. keep=True leaves it in place for further calls of the same step
    (ConvGRU cells)."""
    w = conv.__dict__.get('_prepared') if keep else conv.__dict__.pop('_prepared', None)
    if w is not None and conv.__dict__.get('_prepared_epoch') == ARENA.epoch and w.dtype == dtype and w.shape[-1] == pad8(conv.in_channels):
        return w
    return weight_oihw_to_krsc(conv.weight, dtype)


def pad_vec(v, n):
    if v is None or v.numel() == n:
        return v
    return torch.nn.functional.pad(v, (0, n - v.numel()))


# ----------------------------------------------------------------------------------------------------------------------
# dense convolution (implicit GEMM) with optional bias / ReLU-before-BN epilogue and fused BN statistics
# ----------------------------------------------------------------------------------------------------------------------

BN_SMALL_ROWS = int(_os.environ.get('MG_BN_SMALL_ROWS', '1024'))     # layers up to this many rows run BatchNorm as one launch per direction (csrc/norm_act.hip)
# Off by default (round 3, measured on the frozen workload): the reduce pass it removes (35 launches, 0.44 ms) is paid back by the slower
# data-gradient epilogues (+0.17 ms over 30 launches: two more HBM tiles per output tile at one workgroup per CU) and the replica reduction
# in front of the apply pass (+0.3 ms) -- 13.42 / 13.60 ms with the link against 13.43 / 13.44 ms without. Kept for the tests and as the
# starting point of a register-staged variant; MAGGIE_BN_LINK=1 switches it on.
# Round 5: re-built in a deterministic form and measured again. The epilogue's sums arrive as one row per output tile (mg_conv_stat_rows); ONE
# ordered-sum launch (mg_stat_rows_sum -- the launch that follows bn_bwd_reduce anyway) folds them into [2C] and the apply pass reads that single
# row, so the bn_bwd_reduce kernel (10 us, three tensor reads) is gone for every linked layer; operand-path layers (LazyAct) link too, their
# activation mask re-formed from x * scale + shift inside the epilogue (mg_conv_params.bnb_scale / bnb_shift: only the raw x tile is loaded).
# Same-lease A/B, 2 x 150 steps each: 11.22 / 11.25 ms linked against 11.18 / 11.11 ms unlinked -- the data-gradient kernels are bound by their
# prologue / epilogue latency at this batch size, and a longer epilogue costs what the removed launch saved. Still off by default; parity-tested
# both ways (tests/test_gpu_kernels.py, tests/test_gpu_determinism.py run with either setting).
BN_LINK = _os.environ.get('MAGGIE_BN_LINK', '0') != '0'          # round 6, re-measured with the halo3 register epilogue + prefetched rows: -0.03 ... -0.15 ms per step over three leases, but the 35 linked data-gradient launches grow by ~4 us (family 1.15 -> 1.29 ms, roofline.frac 0.114 -> 0.104): still off


class BnLink:
    """Hand-over between a training BatchNorm(+activation) layer and the ONE convolution that consumes its output z (round 3).

    Backward of such a pair used to be: conv data-gradient kernel -> dz in HBM -> bn_bwd_reduce (reads dz, z, x) -> bn_bwd_apply (reads them
    again). With a link the data-gradient kernel's epilogue -- which holds the dz tile in registers -- writes g = dz * act'(z) and accumulates
    the layer's two reductions (sum g, sum g * xhat) itself (mg_conv_params.bnb_*), so only the apply pass is left: 59 launches and ~1.5 GB of
    reads per step gone. The producer (`conv_bn_act(..., link_out=True)`) promises that z has no other consumer than one conv2d / conv_bn_act
    call (a skip connection taken back through that conv's `carry` output is fine: its gradient is added inside the same epilogue)."""
    __slots__ = ('x2', 'y', 'pack', 'act', 'C', 'M', 'consumers', 'sums', 'g', 'lazy')

    def __init__(self):
        self.x2 = self.y = self.pack = self.sums = self.g = None
        self.act, self.C, self.M, self.consumers, self.lazy = ACT_NONE, 0, 0, 0, False

    def ready(self):
        return self.x2 is not None and self.consumers == 1

    def clear(self):
        self.x2 = self.y = self.pack = self.sums = self.g = None


# Operand-path BatchNorm (round 5; VERDICT round 4 next #1, north_star "conv + BN/ReLU fusions" in TRAINING): conv -> BN -> activation -> conv used to be
# conv (+ statistics) | finalize | apply (read y, write z) | conv (read z). With LAZY_BN the apply pass and the tensor z disappear: the BatchNorm
# layer stops after the finalize kernel and hands its consumer the RAW conv output together with the folded (scale, shift); the consumer's forward
# and weight-gradient kernels form act(y * scale + shift) on the operand's way into LDS (mg_conv_params.xf_*), and the BatchNorm backward re-forms
# the activation mask from y. What travels between the two layers is a LazyAct, NOT a tensor: only conv2d / conv_bn_act understand it, anything
# else fails loudly instead of silently consuming un-normalised values. MAGGIE_LAZY_BN=0 restores the stored form.
LAZY_BN = _os.environ.get('MAGGIE_LAZY_BN', '1') != '0'
LAZY_BN_SYNC = os.environ.get('MAGGIE_LAZY_BN_SYNC', '1') != '0'   # the operand path under SyncBatchNorm too (mailbox exchange only); 0: stored form (A/B)


class _Materialize(torch.autograd.Function):
    """z = act(y * scale + shift) written out (a consumer that cannot transform its operand). The incoming gradient IS dz, which is what
    BNLazy.backward expects from whoever consumed it: identity backward."""

    @staticmethod
    def forward(ctx, t, scale, shift, act, slope):
        C = t.shape[-1]
        return K.affine_act(t.contiguous().view(-1, C), scale, shift, act=act, slope=slope).view(t.shape)

    @staticmethod
    def backward(ctx, dz):
        return dz, None, None, None, None


class LazyAct:
    """A training BatchNorm(+activation) output that was never stored: `t` (the raw conv output, carrying the autograd edge into BNLazy) and the
    transform act(t * scale + shift). Gradients sent back through `t` are gradients with respect to the NORMALISED activation."""
    __slots__ = ('t', 'scale', 'shift', 'act', 'slope', '_z', 'link')

    def __init__(self, t, scale, shift, act, slope, link=None):
        self.t, self.scale, self.shift, self.act, self.slope, self._z, self.link = t, scale, shift, act, slope, None, link

    shape = property(lambda self: self.t.shape)
    dtype = property(lambda self: self.t.dtype)
    device = property(lambda self: self.t.device)
    requires_grad = property(lambda self: self.t.requires_grad)

    def materialize(self):
        if self._z is None:
            self._z = _Materialize.apply(self.t, self.scale, self.shift, self.act, self.slope)
        return self._z


def _linked_sums(rows):
    """Backward sums a data-gradient epilogue left as rows [nrow][2C] -> what the apply pass reads: the rows themselves while they are the few
    replicas of the atomic mode, ONE row after the ordered sum (deterministic mode: one row per output tile, up to tens of thousands)."""
    if rows.dim() == 2 and rows.shape[0] > K.STAT_REPLICAS:
        return K.stat_rows_sum(rows).view(1, -1)
    return rows


class BNLazy(torch.autograd.Function):
    """Training BatchNorm whose apply pass is left to the consumer: batch statistics -> (scale, shift, mean, invstd) + running-stat update, ONE
    launch (mg_bn_finalize over the rows the producing conv's epilogue filled). Returns (alias of x, scale, shift). Backward receives dz -- the
    gradient with respect to act(BN(x)) -- and runs the ordinary reduce + apply pair with the activation mask re-formed from x."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, act, stats, mask_x_pos, link=None, group=None):
        C = x.shape[-1]
        x2 = x.contiguous().view(-1, C)
        M = x2.shape[0]
        centered = False
        if stats is None:
            stats = K.colstats(x2)
        elif stats.dim() == 1 and group is None:                  # (MAGGIE_DETERMINISTIC=0: a sums-only row from the conv epilogue -> exact two-pass variance)
            stats, centered = K.colstats_centered(x2, stats, have_sum=True), True
        cnt_t = None
        if group is not None:
            # SyncBatchNorm (lazy_bn_ok: only with the mailbox exchange): this rank's statistics rows -> [ordered row sum] -> exchange + finalize in
            # ONE launch (mailbox.MailboxComm.bn_finalize) -> the same (scale | shift | mean | invstd) pack, over the rows of ALL ranks
            from . import parallel as _par
            if stats.dim() == 2 and stats.shape[0] > K.STAT_REPLICAS:
                stats = K.stat_rows_sum(stats)
            sc, sh, mean, invstd, cnt_t = _par.SYNCBN_COMM.bn_finalize(stats.contiguous(), float(M), gamma, beta, running_mean, running_var, momentum, eps)
            link = None                                           # (the linked backward sums are local: not under SyncBatchNorm)
        else:
            sc, sh, mean, invstd = K.bn_finalize(stats, M, gamma, beta, running_mean, running_var, momentum, eps, centered=centered)
        pack = sc._base if sc._base is not None else torch.cat([sc, sh, mean, invstd])       # (4, C) = scale | shift | mean | invstd
        ctx.save_for_backward(x2, pack, cnt_t)
        ctx.group = group
        ctx.meta = (x.shape, M, C, act, mask_x_pos)
        ctx.link = None
        if link is not None and (C * x2.element_size()) % 16 == 0 and ((C * x2.element_size()) // 16 & ((C * x2.element_size()) // 16 - 1)) == 0:
            # the consumer's data-gradient epilogue may produce g = dz * act' and this layer's two backward sums (BnLink, lazy form)
            link.x2, link.y, link.pack, link.act, link.C, link.M, link.lazy = x2, None, pack.view(-1), act, C, M, True
            ctx.link = link
        ctx.mark_non_differentiable(sc, sh)
        ctx.set_materialize_grads(False)                          # (else autograd fills a zero gradient for scale and shift on every backward: two launches per layer)
        return x.view_as(x), sc, sh

    @staticmethod
    def backward(ctx, dz, _dsc, _dsh):
        x2, pack, cnt_t = ctx.saved_tensors
        shape, M, C, act, mask_x_pos = ctx.meta
        link, ctx.link = ctx.link, None
        if dz is None:
            return (None,) * 12
        if ctx.group is not None:
            # SyncBatchNorm: dx needs the two sums over ALL ranks' rows, dgamma / dbeta stay local (BNAct.backward): reduce | exchange | apply
            from .parallel import syncbn_exchange_backward
            flat = pack.view(-1)
            sc, sh, mean, invstd = flat[:C], flat[C:2 * C], flat[2 * C:3 * C], flat[3 * C:4 * C]
            dz2 = rows_of(dz, C)
            _, _, local = K.bn_backward(dz2, None, x2, sc, mean, invstd, M, act=act, slope=LRELU_SLOPE, reduce_only=True, mask_x_pos=mask_x_pos,
                                        sums=ARENA.take(2 * C, dz.device), shift=sh)
            sums, local = syncbn_exchange_backward(local, ctx.group)
            dx, _, _ = K.bn_backward(dz2, None, x2, sc, mean, invstd, M, act=act, slope=LRELU_SLOPE, mask_x_pos=mask_x_pos, sums=sums, apply_only=True,
                                     count_ptr=cnt_t, shift=sh)
            if not torch.cuda.is_current_stream_capturing():
                local = local.clone()                             # the arena slice dies with the step; inside a graph it is the graph's own
            return dx.view(shape), local[C:2 * C], local[:C], None, None, None, None, None, None, None, None, None
        if link is not None and link.g is not None and link.sums is not None and link.g.data_ptr() == dz.data_ptr() and dz.is_contiguous():
            # the consumer conv's data-gradient epilogue already wrote g = dz * act'(z) (into `dz`) and one row of the two sums per output tile
            dx, sums = K.bn_bwd_apply_linked(dz.view(-1, C), x2, pack.view(-1), _linked_sums(link.sums), M, mask_x_pos)
            link.clear()
            return dx.view(shape), sums[C:], sums[:C], None, None, None, None, None, None, None, None, None
        if link is not None:
            link.clear()
        sums = ARENA.take(2 * C, dz.device) if torch.cuda.is_current_stream_capturing() else None
        dx, _, sums = K.bn_train_bwd(rows_of(dz, C), None, x2, pack.view(-1), act, LRELU_SLOPE, False, mask_x_pos, sums)
        return dx.view(shape), sums[C:], sums[:C], None, None, None, None, None, None, None, None, None


def lazy_bn_ok(x, bn, res, res2):
    """May this conv -> BatchNorm (+ activation) layer leave its apply pass to the consumer? Training with local batch statistics, 16-bit storage,
    more rows than the one-launch small-layer path takes, no residual entering the activation."""
    C = x.shape[-1]
    if _sync_group(bn) is not None:
        # SyncBatchNorm: with the in-graph mailbox exchange (one launch: exchange + finalize); the collective-based exchanges keep the stored form
        from . import parallel as _par
        comm = _par.SYNCBN_COMM
        if not (LAZY_BN_SYNC and comm is not None and hasattr(comm, 'bn_finalize') and comm.can_finalize(C) and x.is_cuda):
            return False
    return (LAZY_BN and bn.training and torch.is_grad_enabled() and res is None and res2 is None and x.dtype != torch.float32
            and bn.weight is not None and bn.weight.dtype == torch.float32 and bn.weight.numel() == C and bn.running_mean is not None
            and bn.running_mean.numel() == C and x.numel() // C > BN_SMALL_ROWS)


class ConvRaw(torch.autograd.Function):
    """y = [relu]( conv(x, w) + bias ).  x: (N,H,W,Cin) NHWC; w: (Cout, R*S, Cin) KRSC; transposed=True is
    ConvTranspose2d(k, stride, pad). `stats` (fp32 [2*Cout(+1)], zeroed) receives the BN batch statistics of y."""

    @staticmethod
    def forward(ctx, x, w, bias, R, S, stride, pad, dil, transposed, pre_relu, stats, carry=False, link=None, mask_upstream=False, xf=None):
        # xf = (scale, shift, act, slope): x is the RAW output of the producing conv; the BatchNorm + activation between the two layers is applied
        # by the kernels on the operand's way into LDS (forward and weight gradient) -- the normalised activation is never stored (LazyAct)
        ctx.xf = xf
        N, H, W_, Cin = x.shape
        Cout = w.shape[0]
        x = x.contiguous()
        ctx.link, ctx.mask_upstream = link, bool(mask_upstream and pre_relu)
        mode = MODE_TCONV if transposed else MODE_CONV
        Ho = K.conv_out_size(mode, H, R, stride, pad, dil)
        Wo = K.conv_out_size(mode, W_, S, stride, pad, dil)
        ctx.cin_real = getattr(w, '_mg_cin', None)
        y = K.conv_fprop(x.view(-1, Cin), w, mode=mode, N=N, Hin=H, Win=W_, Hout=Ho, Wout=Wo, R=R, S=S, stride=stride,
                         pad=pad, dil=dil, shift=bias, act=ACT_RELU if pre_relu else ACT_NONE, pre_act=False,
                         stats=stats, alg_cin=ctx.cin_real, xf=xf)
        y = y.view(N, Ho, Wo, Cout)
        ctx.wt = getattr(w, '_mg_wt', None)                       # pre-transposed weights from the batched SpectralNorm kernel
        ctx.side = SIDE_WGRAD and getattr(w, '_mg_side_wgrad', False) and not transposed
        ctx.can_park = not (transposed or ctx.side)
        # every use counts; only the plain form below ever parks (alone, or with the other uses of the same weight: _UseGroup)
        ctx.uses = _count_use(w, (N, H, W_, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, x.dtype, ctx.cin_real) if ctx.can_park else None,
                              needs_grad=ctx.needs_input_grad[1])
        ctx.group = getattr(w, '_mg_group', None) if ctx.uses is not None else None
        # mask_upstream: the BatchNorm behind this conv's ReLU applies the ReLU mask in its own backward pass (mask_x_pos), y is not needed
        ctx.save_for_backward(x, w, y if (pre_relu and not ctx.mask_upstream) else None)
        ctx.geom = (N, H, W_, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, transposed, pre_relu, bias is not None)
        if carry:
            # `carry`: the input is handed back as a second output for the caller's skip connection. Its gradient then arrives HERE (d_carry)
            # and is added inside the data-gradient kernel's epilogue (res2) -- instead of autograd summing the two branches of x with a
            # separate feature-map-sized add kernel per residual block.
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, d_carry=None):
        x, w, y = ctx.saved_tensors
        N, H, W_, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, transposed, pre_relu, has_bias = ctx.geom
        dy2 = dy.contiguous().view(-1, Cout)
        dx = dw = db = None
        want_db = has_bias and ctx.needs_input_grad[2]
        mask_here = pre_relu and not ctx.mask_upstream
        if mask_here or want_db:                                  # ReLU mask and bias gradient in one pass
            dy2, db = K.bias_act_bwd(dy2, y.view(-1, Cout) if mask_here else None, want_db)
        if ctx.needs_input_grad[0]:
            wt = ctx.wt if ctx.wt is not None else w.permute(2, 1, 0).contiguous()      # (Cin, taps, Cout)
            dmode = MODE_CONV if transposed else MODE_TCONV
            r2 = None if d_carry is None else d_carry.to(dy2.dtype).contiguous().view(-1, Cin)
            link, bnb, sums_rep = ctx.link, None, None
            if link is not None and link.ready() and link.C == Cin and link.M == N * H * W_ and link.x2.dtype == dy2.dtype:
                # x is the output of a training BatchNorm layer with no other consumer: its backward reductions ride on this kernel's epilogue
                Cb = link.C
                nrow = K.conv_stat_rows(N * H * W_, N, H, W_)       # one row per output tile in deterministic mode, else 32 replicas
                sums_rep = ARENA.take(nrow * 2 * Cb, dy2.device).view(nrow, 2 * Cb)
                if link.lazy:                                   # the layer never stored its activation output: mask from x * scale + shift
                    bnb = (None, link.x2, link.pack[2 * Cb:3 * Cb], link.pack[3 * Cb:4 * Cb], link.act, link.pack[:Cb], link.pack[Cb:2 * Cb])
                else:
                    bnb = (link.y if link.act != ACT_NONE else None, link.x2, link.pack[2 * Cb:3 * Cb], link.pack[3 * Cb:4 * Cb], link.act)
            dx = K.conv_fprop(dy2, wt, mode=dmode, N=N, Hin=Ho, Win=Wo, Hout=H, Wout=W_, R=R, S=S, stride=stride, pad=pad,
                              dil=dil, alg_cout=ctx.cin_real, res2=r2, stats=sums_rep, bnb=bnb, slope=LRELU_SLOPE).view(N, H, W_, Cin)
            if bnb is not None:
                link.sums, link.g = sums_rep, dx
        elif d_carry is not None:
            dx = d_carry
        if ctx.needs_input_grad[1]:
            if ctx.side:
                side, main = fork_side(dy2.device)
                with torch.cuda.stream(side):
                    dw = K.conv_wgrad(x.view(-1, Cin), dy2, cout=Cout, mode=MODE_CONV, N=N, Hin=H, Win=W_, Hout=Ho, Wout=Wo,
                                      R=R, S=S, stride=stride, pad=pad, dil=dil, out_dtype=w.dtype)
                x.record_stream(side)                             # allocator: not to be recycled under the side stream's feet
                dy2.record_stream(side)
                dw.record_stream(main)
            elif not transposed:
                def wgrad(out, park, park_ws):
                    return K.conv_wgrad(x.view(-1, Cin), dy2, cout=Cout, mode=MODE_CONV, N=N, Hin=H, Win=W_, Hout=Ho, Wout=Wo, R=R, S=S, stride=stride,
                                        pad=pad, dil=dil, out=out, out_dtype=w.dtype, alg_cin=ctx.cin_real, park=park, xf=ctx.xf, park_ws=park_ws)
                single = ctx.can_park and ctx.uses is not None and ctx.uses[0] == 1
                if not single and GROUP_WGRAD and ctx.can_park and ctx.group is not None and ctx.group.uniform():
                    dw = _group_wgrad(ctx.group, wgrad)           # several calls, one geometry: one reduction over all their slabs
                else:
                    dw = wgrad(None, _park_list() if single else None, None)
            else:
                # dW[ci, tap, co] = sum_o x[o, ci] * dy[2o - pad + k, co]  (roles of x and dy swapped)
                dwt = K.conv_wgrad(dy2, x.view(-1, Cin), cout=Cin, mode=MODE_CONV, N=N, Hin=Ho, Win=Wo, Hout=H, Wout=W_,
                                   R=R, S=S, stride=stride, pad=pad, dil=dil, out_dtype=w.dtype)
                # the batched weight pipeline's backward reads this gradient in the layout it was written in (a permuted view, no copy)
                dw = dwt.permute(2, 1, 0) if getattr(w, '_mg_join', False) else dwt.permute(2, 1, 0).contiguous()
        return dx, dw, db, None, None, None, None, None, None, None, None, None, None, None, None


def conv2d(x, w, bias=None, R=3, S=3, stride=1, pad=1, dil=1, transposed=False, pre_relu=False, stats=None, carry=False, mask_upstream=False):
    xf = None
    if isinstance(x, LazyAct):
        # the producer left BatchNorm + activation to this convolution: taken when BOTH kernel forms this geometry dispatches to (forward, weight
        # gradient) transform their operand in flight and nothing else needs the normalised tensor (a `carry` skip connection would); otherwise
        # it is written out here, once, by the ordinary apply pass
        N_, H_, W__, Cin_ = x.shape
        ok = not (transposed or carry or SIDE_WGRAD) and x.t.is_contiguous() and \
            K.conv_xform_ok(x.dtype, N_, H_, W__, Cin_, w.shape[0], R, S, stride, pad, dil, 0) and \
            (not w.requires_grad or K.conv_xform_ok(x.dtype, N_, H_, W__, Cin_, w.shape[0], R, S, stride, pad, dil, 1))
        if ok:
            xf, lz, x = (x.scale, x.shift, x.act, x.slope), x, x.t
            if lz.link is not None:
                x._mg_bnlink = lz.link
        else:
            x = x.materialize()
    link = getattr(x, '_mg_bnlink', None) if (BN_LINK and torch.is_grad_enabled()) else None
    if link is not None:
        link.consumers += 1                                       # a link seen by two convolutions is void (BnLink.ready)
    return ConvRaw.apply(x, w, bias, R, S, stride, pad, dil, transposed, pre_relu, stats, carry, link, mask_upstream, xf)


def linear_rows(x2d, w, bias=None, pre_relu=False, stats=None):
    """Per-row linear map (1x1 conv over a rows x channels matrix). w: (Cout, 1, Cin)."""
    R_, C = x2d.shape
    return conv2d(x2d.view(1, 1, R_, C), w, bias, 1, 1, 1, 0, 1, False, pre_relu, stats).view(R_, -1)


# ----------------------------------------------------------------------------------------------------------------------
# BatchNorm (+ residual + activation), training (batch statistics, running-stat update, SyncBN) and eval
# ----------------------------------------------------------------------------------------------------------------------

SYNCBN_WORLD1 = os.environ.get('MAGGIE_SYNCBN_WORLD1', '0') == '1'   # test hook: run the SyncBN exchange code also in a 1-rank process group


def _sync_group(bn):
    if isinstance(bn, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or SYNCBN_WORLD1):
        return bn.process_group if bn.process_group is not None else dist.group.WORLD
    return None


def rows_of(t, C):
    """(rows, C) view of a channels-last tensor for the row-wise kernels (they take a row pitch): no copy when `t` is a channel slice of a wider
    row-major buffer -- the gradient torch.cat hands to each of its inputs (the five ASPP branches) -- else the contiguous copy."""
    if t.is_contiguous():
        return t.view(-1, C)
    if t.dim() >= 2 and t.shape[-1] == C and t.stride(-1) == 1 and t.data_ptr() % 16 == 0:
        ld = t.stride(-2)
        ok, pitch = ld >= C and ld % 8 == 0, ld
        for d in range(t.dim() - 2, -1, -1):                      # the leading dimensions must walk rows of one pitch, row-major
            if t.shape[d] > 1 and t.stride(d) != pitch:
                ok = False
            pitch *= t.shape[d]
        if ok:
            return t.as_strided((t.numel() // C, C), (ld, 1))
    return t.contiguous().view(-1, C)


class BNAct(torch.autograd.Function):
    """y = act( BN(x) + res ).  x, y: (..., C) rows x channels. Training uses batch statistics (optionally
    pre-accumulated by the producing conv's epilogue) and updates running stats in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, running_mean, running_var, training, momentum, eps, act, stats, res_mode, group,
                mask_x_pos, link=None, count_mult=1):
        ctx.link = None
        shape = x.shape
        C = shape[-1]
        x2 = x.contiguous().view(-1, C)
        M = x2.shape[0]
        H, W_ = (shape[1], shape[2]) if x.dim() == 4 else (1, 1)
        if training and group is None and gamma.dtype == torch.float32 and gamma.numel() == C and M > 0 and \
                (running_mean is None or running_mean.numel() == C):
            # the common case (local statistics, unpadded channels) in one C call: statistics, finalize, apply
            # exact two-pass variance: <= EXACT_STATS_ROWS rows. Deterministic mode keeps it for the <= BN_SMALL_ROWS layers (one workgroup, no
            # cross-workgroup step) and for fp32 storage (ordered two-pass form, csrc/norm_act.hip: with 16-bit storage the rounding of x itself,
            # 2^-8 |x|, is far above what E[x^2] - E[x]^2 loses; with fp32 storage it is not -- ADVICE round 4)
            exact = M <= BN_SMALL_ROWS or (M <= EXACT_STATS_ROWS and (not K.hip.DETERMINISTIC or x2.dtype == torch.float32))
            if exact and stats is not None and stats.dim() != 1:
                stats = None                                      # replicas from a conv epilogue: the exact path recomputes
            r2 = None if res is None else res.contiguous().view(-1, C)
            own = stats is None
            small = exact and 1 < M <= BN_SMALL_ROWS              # one workgroup per channel chunk: no statistics scratch at all
            ws = ARENA.take(K.stats_ws_floats(C, exact), x.device) if (own and not small) else None      # zeroed once per step / per graph
            y, pack = K.bn_train_fwd(x2, gamma, beta, running_mean, running_var, momentum, eps, act, LRELU_SLOPE, r2, res_mode, H, W_,
                                     stats, exact, ws, count_mult=count_mult)
            ctx.save_for_backward(x2, y, pack)
            ctx.fast = True
            ctx.meta = (shape, M, C, act, res is not None, res_mode, training, group, mask_x_pos, C, res.shape if res is not None else None)
            if link is not None and M > BN_SMALL_ROWS and (C * x2.element_size()) % 16 == 0 and ((C * x2.element_size()) // 16 & ((C * x2.element_size()) // 16 - 1)) == 0:
                # (layers up to BN_SMALL_ROWS rows already run their backward as ONE launch; the linked apply kernel wants a power-of-two
                # number of 16-byte chunks per row)
                link.x2, link.y, link.pack, link.act, link.C, link.M = x2, y, pack, act, C, M
                ctx.link = link
            return y.view(shape)
        ctx.fast = False
        g32 = pad_vec(gamma.float(), C)
        b32 = pad_vec(beta.float(), C)
        count = float(M)
        cnt_t = None
        centered = False
        if training:
            if group is None and M <= EXACT_STATS_ROWS:
                # few samples per channel: exact two-pass variance (E[x^2]-E[x]^2 cancels badly when var << mean^2). A 1-D
                # `stats` row already holds the column sums from the producing conv's epilogue (new_stats(.., rows)).
                if stats is not None and stats.dim() == 1:
                    stats, centered = K.colstats_centered(x2, stats, have_sum=True), True
                else:
                    stats, centered = K.colstats_centered(x2, ARENA.take(2 * C, x.device)), True
            elif stats is None:
                stats = K.colstats(x2)
            fused_comm = None
            if group is not None:
                from . import parallel as _par
                comm = _par.SYNCBN_COMM
                if comm is not None and hasattr(comm, 'bn_finalize') and comm.can_finalize(C) and x.is_cuda and stats.shape[-1] == 2 * C:
                    fused_comm = comm                             # mailbox exchange: replica sums, exchange and finalize in ONE launch (below)
                    if stats.dim() == 2 and stats.shape[0] > K.STAT_REPLICAS:
                        stats = K.stat_rows_sum(stats)            # one row per tile / row block (deterministic mode): added in order, in parallel, first
                else:
                    flat = (K.stat_rows_sum(stats) if x.is_cuda else stats.sum(0)) if stats.dim() == 2 else stats
                    pack = _par.syncbn_exchange_forward(torch.cat([flat[:2 * C], torch.full((1,), float(M), device=x.device)]), group)
                    stats, cnt_t = pack[:2 * C], pack[2 * C:]
            rm = running_mean if running_mean.numel() == C else None
            rv = running_var if running_var.numel() == C else None
            rm_p, rv_p = rm, rv
            if rm is None and running_mean is not None:       # padded channel count: update through a temp
                rm_p, rv_p = pad_vec(running_mean, C).clone(), pad_vec(running_var, C).clone()
            if fused_comm is not None:
                scale, shift, mean, invstd, cnt_t = fused_comm.bn_finalize(stats.contiguous(), count, g32, b32, rm_p, rv_p, momentum, eps)
            else:
                scale, shift, mean, invstd = K.bn_finalize(stats, count, g32, b32, rm_p, rv_p, momentum, eps, count_ptr=cnt_t, centered=centered)
            if rm is None and running_mean is not None:
                running_mean.copy_(rm_p[:running_mean.numel()])
                running_var.copy_(rv_p[:running_var.numel()])
        else:
            scale, shift = K.bn_fold(g32, b32, pad_vec(running_mean, C), pad_vec(running_var, C), eps)
            mean = pad_vec(running_mean, C)
            invstd = torch.rsqrt(pad_vec(running_var, C) + eps)
        r2 = None if res is None else res.contiguous().view(-1, C)
        y = K.affine_act(x2, scale, shift, res=r2, res_mode=res_mode, act=act, slope=LRELU_SLOPE, H=H, W=W_)
        ctx.save_for_backward(x2, y, scale, mean, invstd, cnt_t)
        ctx.meta = (shape, M, C, act, res is not None, res_mode, training, group, mask_x_pos, gamma.numel(), res.shape if res is not None else None)
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        if ctx.fast:
            x2, y, pack = ctx.saved_tensors
            shape, M, C, act, has_res, res_mode, training, group, mask_x_pos, nch, res_shape = ctx.meta
            link = ctx.link
            if link is not None and link.g is not None and link.sums is not None and link.g.data_ptr() == dy.data_ptr() and dy.is_contiguous():
                # the consumer conv's data-gradient epilogue already produced g = dy * act'(y) (in `dy`) and the two reductions
                dx, sums = K.bn_bwd_apply_linked(dy.view(-1, C), x2, pack, _linked_sums(link.sums), M, mask_x_pos)
                dres = dy.view(-1, C) if has_res else None
                link.clear()
                if has_res:
                    dres = K.pool2x2(dres, 1, shape[0], shape[1] // 2, shape[2] // 2).view(res_shape) if res_mode == 2 else dres.view(res_shape)
                return dx.view(shape), sums[C:], sums[:C], dres, None, None, None, None, None, None, None, None, None, None, None, None
            if link is not None:
                link.clear()
            # inside a graph capture the accumulator is a slice of the graph's own zero arena (no fill kernel per layer); eagerly the
            # sums leave as gradients and must outlive the step's arena, so they get their own (zeroed in the call)
            sums = ARENA.take(2 * C, dy.device) if torch.cuda.is_current_stream_capturing() else None
            dx, dres, sums = K.bn_train_bwd(rows_of(dy, C), y, x2, pack, act, LRELU_SLOPE, has_res, mask_x_pos, sums)
            if has_res:
                dres = K.pool2x2(dres, 1, shape[0], shape[1] // 2, shape[2] // 2).view(res_shape) if res_mode == 2 else dres.view(res_shape)
            return dx.view(shape), sums[C:], sums[:C], dres, None, None, None, None, None, None, None, None, None, None, None, None
        x2, y, scale, mean, invstd, cnt_t = ctx.saved_tensors
        shape, M, C, act, has_res, res_mode, training, group, mask_x_pos, nch, res_shape = ctx.meta
        dy2 = dy.contiguous().view(-1, C)
        if training:
            _, _, sums = K.bn_backward(dy2, y, x2, scale, mean, invstd, M, act=act, slope=LRELU_SLOPE, reduce_only=True,
                                       sums=ARENA.take(2 * C, dy.device))
            local = sums
            if group is not None:
                # SyncBN: dx needs the sums over ALL ranks' rows; dgamma / dbeta stay LOCAL (nn.SyncBatchNorm semantics -- the
                # data-parallel gradient averaging that follows would otherwise count them world_size times)
                from .parallel import syncbn_exchange_backward
                sums, local = syncbn_exchange_backward(sums, group)
            dx, dres, _ = K.bn_backward(dy2, y, x2, scale, mean, invstd, M, act=act, slope=LRELU_SLOPE, want_dres=has_res,
                                        mask_x_pos=mask_x_pos, sums=sums, apply_only=True, count_ptr=cnt_t)
            if not torch.cuda.is_current_stream_capturing():
                local = local.clone()            # the arena slice dies with the step; inside a graph it is the graph's own
            dgamma, dbeta = local[C:2 * C][:nch], local[:C][:nch]
        else:
            # eval statistics are constants: dx = g * scale
            zeros = torch.zeros(2 * C, dtype=torch.float32, device=dy.device)
            dx, dres, _ = K.bn_backward(dy2, y, x2, scale, mean, invstd, M, act=act, slope=LRELU_SLOPE, want_dres=has_res,
                                        mask_x_pos=mask_x_pos, sums=zeros, apply_only=True)
            _, _, sums = K.bn_backward(dy2, y, x2, scale, mean, invstd, M, act=act, slope=LRELU_SLOPE, reduce_only=True)
            dgamma, dbeta = sums[C:2 * C][:nch].clone(), sums[:C][:nch].clone()
        dx = dx.view(shape)
        if has_res:
            if res_mode == 2:
                N, H, W_ = shape[0], shape[1], shape[2]
                dres = K.pool2x2(dres, 1, N, H // 2, W_ // 2).view(res_shape)
            else:
                dres = dres.view(res_shape)
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None, None, None, None, None


def batch_norm_act(x, bn, act=ACT_NONE, res=None, stats=None, res_mode=1, mask_x_pos=False, link=None, count_mult=1):
    """`bn` is an nn.BatchNorm{1,2}d / nn.SyncBatchNorm used as the parameter + running-stat holder."""
    training = bn.training or (bn.running_mean is None)
    if training and bn.num_batches_tracked is not None and not DEFER_BN_COUNTERS:
        if BN_COUNT_LOG is not None:
            BN_COUNT_LOG.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    mom = 0.1 if bn.momentum is None else bn.momentum
    return BNAct.apply(x, bn.weight, bn.bias, res, bn.running_mean, bn.running_var, training, mom, bn.eps, act, stats, res_mode,
                       _sync_group(bn) if training else None, mask_x_pos, link, count_mult)


def new_stats(channels, device, rows=None, bn=None, geom=None, dtype=None):
    """Zeroed accumulator for the conv epilogue's BatchNorm statistics. Layers that will use the exact two-pass variance
    (`rows` <= EXACT_STATS_ROWS, no SyncBN) only need the column sums: one row [2*channels] (conv stat_mode 1)."""
    if K.hip.DETERMINISTIC:
        if rows is not None and (rows <= BN_SMALL_ROWS or (rows <= EXACT_STATS_ROWS and dtype == torch.float32)) and (bn is None or _sync_group(bn) is None):
            return None                                            # the one-workgroup / ordered two-pass BatchNorm computes its own (exact) statistics
        n = K.conv_stat_rows(rows if rows is not None else 1, *geom) if geom else K.conv_stat_rows(rows if rows is not None else 1)
        return ARENA.take(n * 2 * channels, device).view(n, 2 * channels)
    if rows is not None and rows <= EXACT_STATS_ROWS and (bn is None or _sync_group(bn) is None):
        return ARENA.take(2 * channels, device)
    return ARENA.take(K.STAT_REPLICAS * 2 * channels, device).view(K.STAT_REPLICAS, 2 * channels)


def conv_bn_act(x, w, bn, act=ACT_NONE, R=3, S=3, stride=1, pad=1, dil=1, transposed=False, res=None, res_mode=1, res2=None,
                relu_before_bn=False, bias=None, carry=False, link_out=False, count_mult=1, lazy_out=False):
    """conv -> BN -> (+res) -> act (-> +res2).  In inference (no grad, eval BN) this is ONE fused kernel; in training the
    conv epilogue accumulates the batch statistics and a second HBM pass applies them.
    `link_out`: the caller guarantees that the returned activation is consumed by exactly ONE conv2d / conv_bn_act call (see BnLink).
    `lazy_out`: same guarantee, and the caller passes the result ONLY to conv2d / conv_bn_act: in training it is then a LazyAct (raw conv output +
    folded batch statistics; the consumer's kernels apply BatchNorm + activation to their operand in flight) instead of a tensor.
    `x` itself may be a LazyAct."""
    if isinstance(x, LazyAct) and (carry or ((not bn.training) and not torch.is_grad_enabled())):
        x = x.materialize()
    Cout = w.shape[0]
    fused = (not bn.training) and (not torch.is_grad_enabled())
    if fused:
        N, H, W_, Cin = x.shape
        mode = MODE_TCONV if transposed else MODE_CONV
        Ho = K.conv_out_size(mode, H, R, stride, pad, dil)
        Wo = K.conv_out_size(mode, W_, S, stride, pad, dil)
        scale, shift = K.bn_fold(pad_vec(bn.weight.float(), Cout), pad_vec(bn.bias.float(), Cout), pad_vec(bn.running_mean, Cout),
                                 pad_vec(bn.running_var, Cout), bn.eps)
        if bias is not None:
            shift = shift + scale * bias if not relu_before_bn else shift
        y = K.conv_fprop(x.contiguous().view(-1, Cin), w, mode=mode, N=N, Hin=H, Win=W_, Hout=Ho, Wout=Wo, R=R, S=S, stride=stride,
                         pad=pad, dil=dil, scale=scale, shift=shift,
                         res=None if res is None else res.contiguous().view(-1, Cout), res_mode=res_mode,
                         res2=None if res2 is None else res2.contiguous().view(-1, Cout),
                         act=ACT_RELU if relu_before_bn else act, pre_act=relu_before_bn, slope=LRELU_SLOPE)
        return (y.view(N, Ho, Wo, Cout), x) if carry else y.view(N, Ho, Wo, Cout)
    stats = None
    if bn.training:
        mode_ = MODE_TCONV if transposed else MODE_CONV
        ho_, wo_ = K.conv_out_size(mode_, x.shape[1], R, stride, pad, dil), K.conv_out_size(mode_, x.shape[2], S, stride, pad, dil)
        rows = x.shape[0] * ho_ * wo_
        # fused statistics in the conv epilogue -- except on the largest, thinnest tensors (UNFUSED_STATS_ROWS)
        stats = new_stats(Cout, x.device, rows, bn, geom=(x.shape[0], ho_, wo_), dtype=x.dtype) if rows < UNFUSED_STATS_ROWS else None
    xc = None
    # ReLU-before-BN (encoder shortcuts): the BatchNorm's backward applies the ReLU mask itself (mask_x_pos: its input IS the ReLU output),
    # so the conv's backward needs neither its saved output nor a masking pass over the gradient
    mask_up = bool(relu_before_bn and bn.training and bias is None)
    if carry and torch.is_grad_enabled() and x.requires_grad:
        y, xc = conv2d(x, w, bias, R, S, stride, pad, dil, transposed, relu_before_bn, stats, True, mask_upstream=mask_up)
    else:
        y = conv2d(x, w, bias, R, S, stride, pad, dil, transposed, relu_before_bn, stats, mask_upstream=mask_up)
    if lazy_out and count_mult == 1 and bias is None and lazy_bn_ok(y, bn, res, res2):
        if bn.num_batches_tracked is not None and not DEFER_BN_COUNTERS:
            if BN_COUNT_LOG is not None:
                BN_COUNT_LOG.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked.add_(1)
        a = ACT_NONE if relu_before_bn else act
        lk = BnLink() if BN_LINK else None
        t, sc, sh = BNLazy.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, 0.1 if bn.momentum is None else bn.momentum, bn.eps, a, stats, mask_up, lk,
                                 _sync_group(bn))
        lz = LazyAct(t, sc, sh, a, LRELU_SLOPE, lk if (lk is not None and lk.x2 is not None) else None)
        return (lz, x if xc is None else xc) if carry else lz
    link = BnLink() if (link_out and BN_LINK and bn.training and res2 is None and torch.is_grad_enabled()) else None
    y = batch_norm_act(y, bn, ACT_NONE if relu_before_bn else act, res=res, stats=stats, res_mode=res_mode, mask_x_pos=mask_up, link=link,
                       count_mult=count_mult)
    if link is not None and link.x2 is not None:
        y._mg_bnlink = link
    if res2 is not None:
        y = y + res2
    return (y, x if xc is None else xc) if carry else y


# ----------------------------------------------------------------------------------------------------------------------
# sparse (gather) convolution over active sites
# ----------------------------------------------------------------------------------------------------------------------

class GatherConv(torch.autograd.Function):
    """y[r] = act( sum_k W[:, k, :] x[nbr[r, k]] + bias ).  `nbr_t`: table for the input-gradient gather
    (same table with reversed taps for submanifold convs, the strided table for inverse convs)."""

    @staticmethod
    def forward(ctx, x, w, bias, nbr, nbr_t, reverse_taps, ksize, stats, act):
        Cout = w.shape[0]
        x = x.contiguous()
        y = K.conv_fprop(x, w, mode=MODE_GATHER, nbr=nbr, R=ksize, S=ksize, shift=bias, stats=stats, act=act, pre_act=False)
        ctx.save_for_backward(x, w, nbr, nbr_t, y if act != ACT_NONE else None)
        ctx.meta = (reverse_taps, ksize, Cout, bias is not None, act)
        ctx.wt = getattr(w, '_mg_wt', None)          # (Cin_pad, taps [reversed for submanifold], Cout) twin from the weight bank
        ctx.uses = _count_use(w, needs_grad=ctx.needs_input_grad[1])
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, nbr, nbr_t, y = ctx.saved_tensors
        reverse_taps, ksize, Cout, has_bias, act = ctx.meta
        dy = dy.contiguous()
        dx = dw = db = None
        want_db = has_bias and ctx.needs_input_grad[2]
        if act == ACT_RELU or want_db:
            dy, db = K.bias_act_bwd(dy, y if act == ACT_RELU else None, want_db)
        if ctx.needs_input_grad[0]:
            wt = ctx.wt
            if wt is None:
                wt = w.permute(2, 1, 0)
                if reverse_taps:
                    wt = wt.flip(1)
                wt = wt.contiguous()
            dx = K.conv_fprop(dy, wt, mode=MODE_GATHER, nbr=nbr_t, R=ksize, S=ksize)
        if ctx.needs_input_grad[1]:
            park = _park_list() if (ctx.uses is not None and ctx.uses[0] == 1) else None
            dw = K.conv_wgrad(x, dy, cout=Cout, mode=MODE_GATHER, nbr=nbr, R=ksize, S=ksize, out_dtype=w.dtype, park=park)
        return dx, dw, db, None, None, None, None, None, None


def gather_conv(x, w, nbr, nbr_t, reverse_taps, ksize=3, bias=None, stats=None, act=ACT_NONE):
    return GatherConv.apply(x, w, bias, nbr, nbr_t, reverse_taps, ksize, stats, act)


class BitsSelect(torch.autograd.Function):
    """out = a where the bit plane is set, b elsewhere (the blend a*w + b*(1-w) of `fuse` for a 0/1 weight plane), fp32 planes."""

    @staticmethod
    def forward(ctx, bits, a, b, W):
        a, b = a.float().contiguous(), b.float().contiguous()
        ctx.save_for_backward(bits)
        ctx.W = W
        return K.bits_select(bits, a, b, W)

    @staticmethod
    def backward(ctx, dy):
        (bits,) = ctx.saved_tensors
        da, db = K.bits_select_bwd(bits, dy.contiguous(), ctx.W, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return None, da, db, None


def bits_select(bits, a, b, W):
    return BitsSelect.apply(bits, a, b, W)


class GatherRows(torch.autograd.Function):
    """rows[r] = dense[frame(r), y, x] (* tokens[frame, inst]).  `bits`/`wordoff`: the level's bit planes and ranks (used by the
    atomic-free backward)."""

    @staticmethod
    def forward(ctx, dense, coords, bits, wordoff, n_i, mul):
        dense = dense.contiguous()
        mul32 = None if mul is None else mul.float().contiguous()
        y = K.gather_rows(dense, coords, n_i, mul=mul32)
        ctx.save_for_backward(dense, coords, bits, wordoff, mul32)
        ctx.meta = (n_i, dense.shape, mul is not None, None if mul is None else mul.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        dense, coords, bits, wordoff, mul32 = ctx.saved_tensors
        n_i, dshape, has_mul, mul_dtype = ctx.meta
        dy = dy.contiguous()
        ddense = dmul = None
        if ctx.needs_input_grad[0]:
            ddense = K.gather_rows_bwd_dense(dy, bits, wordoff, n_i, dshape, mul=mul32)
        if has_mul and ctx.needs_input_grad[5]:
            _, dmul = K.gather_rows_bwd(dy, coords, n_i, dshape, mul=mul32, dense=dense, want_ddense=False, want_dmul=True)
            dmul = dmul.to(mul_dtype)
        return ddense, None, None, None, None, dmul


def gather_rows(dense, level, n_i, mul=None):
    """`level`: object with .coords, .bits, .wordoff of the resolution level `dense` lives at."""
    return GatherRows.apply(dense, level.coords, level.bits, level.wordoff, n_i, mul)


class ScatterPlane(torch.autograd.Function):
    """SparseConvTensor.dense() with the reference's -99 background (resnet_inst_matt_spconv.py:247-251,264-268)."""

    @staticmethod
    def forward(ctx, vals, coords, P, H, W, fill):
        plane = K.scatter_plane(vals, 0, coords, P, H, W, fill)
        ctx.save_for_backward(coords)
        ctx.meta = (vals.shape, vals.dtype)
        return plane

    @staticmethod
    def backward(ctx, dplane):
        (coords,) = ctx.saved_tensors
        vshape, vdtype = ctx.meta
        g = K.gather_plane(dplane.contiguous(), coords, vdtype)
        if vshape[1] != 1:
            g = torch.nn.functional.pad(g, (0, vshape[1] - 1))
        return g, None, None, None, None, None


def scatter_plane(vals, coords, P, H, W, fill=-99.0):
    return ScatterPlane.apply(vals, coords, P, H, W, fill)


class UpsampleTanh(torch.autograd.Function):
    """(tanh(bilinear_up(x, scale)) + 1) / 2 -> fp32 planes (N, C, h*scale, w*scale). x: NHWC rows (N,h,w,Cpad) when
    `nhwc`, else fp32 planes (N, C, h, w)."""

    @staticmethod
    def forward(ctx, x, C, scale, nhwc, apply_tanh, pscale=None, want_flag=False):
        x = x.contiguous()
        if nhwc:
            N, h, w, Cp = x.shape
            strides = (h * w * Cp, 1, w * Cp, Cp)
        else:
            N, _, h, w = x.shape
            strides = (C * h * w, h * w, w, 1)
        ps = None if pscale is None else pscale.detach().float().reshape(-1).contiguous()
        flag = ARENA.zeros(1, x.device, torch.int32) if want_flag else None
        out = K.upsample_tanh(x, strides, N, C, h, w, scale, apply_tanh, ps, flag)
        ctx.save_for_backward(out, ps)
        ctx.meta = (x.shape, x.dtype, strides, N, C, h, w, scale, apply_tanh)
        if want_flag:
            ctx.mark_non_differentiable(flag)
            return out, flag
        return out

    @staticmethod
    def backward(ctx, dout, _dflag=None):
        out, ps = ctx.saved_tensors
        xshape, xdtype, strides, N, C, h, w, scale, apply_tanh = ctx.meta
        # scale 1 and the tiled x4 / x8 kernels write every element of the C planes exactly once: only padded NHWC channels (and the atomic
        # fallback of other scales) need a zeroed buffer -- the OS1 planes alone were a 42 MB fill per step
        covers = (scale == 1 or (scale in (4, 8) and h % 8 == 0 and w % 8 == 0)) and (len(xshape) != 4 or strides[1] != 1 or xshape[-1] == C)
        din = (torch.empty if covers else torch.zeros)(xshape, dtype=torch.float32, device=dout.device)
        K.upsample_tanh_bwd(dout.contiguous(), out, strides, N, C, h, w, scale, din, apply_tanh, ps)
        return din.to(xdtype), None, None, None, None, None, None


def upsample_tanh(x, C, scale, nhwc, apply_tanh=True, pscale=None, want_flag=False):
    """`pscale` (N*C values in {0, 1}): per-plane scale of the output (`* valid_masks`); `want_flag`: also return an int32 [1] tensor that is 1
    when any output element is non-zero."""
    return UpsampleTanh.apply(x, C, scale, nhwc, apply_tanh, pscale, want_flag)


class MaskEmbed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, masks, table, dtype):
        out = K.mask_embed(image.contiguous(), masks.contiguous(), table.float().contiguous(), dtype)
        ctx.save_for_backward(masks)
        ctx.tshape = tuple(table.shape)
        return out

    @staticmethod
    def backward(ctx, dx):
        (masks,) = ctx.saved_tensors
        return None, None, K.mask_embed_bwd(dx.contiguous(), masks.contiguous(), ctx.tshape), None


def mask_embed(image, masks, table, dtype):
    return MaskEmbed.apply(image, masks, table, dtype)


class AvgPool2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        N, H, W_, C = x.shape
        ctx.shape = x.shape
        return K.pool2x2(x.contiguous().view(-1, C), 0, N, H // 2, W_ // 2).view(N, H // 2, W_ // 2, C)

    @staticmethod
    def backward(ctx, dy):
        N, H, W_, C = ctx.shape
        return K.pool2x2(dy.contiguous().view(-1, C), 2, N, H, W_).view(N, H, W_, C)


def avg_pool2x2(x):
    return AvgPool2x2.apply(x)


# ----------------------------------------------------------------------------------------------------------------------
# region ops (no gradients: integer / boolean work on bit planes)
# ----------------------------------------------------------------------------------------------------------------------

def draw_widths(P, k_size):
    """The P per-slice dilation widths of compute_unknown(is_train=True) (maggie/utils/utils.py:47: one np.random.randint(1, k) per slice)
    as ONE vectorised draw = the same values and the same generator state afterwards (tests/test_host_cpu.py), a tenth of the host time."""
    import numpy as np
    return np.random.randint(1, k_size, size=P).astype(np.int32)


def unknown_bits(alpha, k_size=30, is_train=False, andmask=None, widths=None):
    """compute_unknown (maggie/utils/utils.py:28-55) on fp32 planes (..., H, W) -> bit planes (P, H, Ww).
    Train mode draws one np.random.randint(1, k_size) per slice from the GLOBAL numpy RNG (reference order) -- unless the caller
    already drew them (`widths`: device int32 [P]; the captured detail stage gets them as a graph input)."""
    import numpy as np
    a = alpha.detach()
    if a.dtype != torch.float32:
        a = a.float()
    a = a.contiguous()
    H, W_ = a.shape[-2:]
    P = a.numel() // (H * W_)
    bits = K.bits_pack(a, mode=0)
    if is_train:
        wd = widths if widths is not None else torch.from_numpy(draw_widths(P, k_size)).to(a.device, non_blocking=True)
        return K.bits_dilate(bits, W_, widths=wd, andmask=andmask)
    return K.bits_dilate(bits, W_, width=k_size // 2, andmask=andmask)


# ----------------------------------------------------------------------------------------------------------------------
# fused matting losses on fp32 planes
# ----------------------------------------------------------------------------------------------------------------------

class MattingLosses(torch.autograd.Function):
    """(weighted L1, Laplacian-pyramid L1, Sobel-gradient L1) of pred vs target under `weight`, all (.., H, W) fp32 planes.
    Returns a 3-vector; backward yields d/dpred only (target and weight are constants of the loss)."""

    @staticmethod
    def forward(ctx, pred, target, weight, pvalid=None):
        H, W_ = pred.shape[-2:]
        assert H % 8 == 0 and W_ % 8 == 0
        p = pred.detach().float().contiguous()
        t = target.detach().float().contiguous()
        w = weight.detach().float().contiguous()
        P = p.numel() // (H * W_)
        dev = p.device
        hipc, c_int = K.hip.call, K.c_int
        ptr, st = K.hip.ptr, K.hip.stream
        flags = ARENA.acc(P, dev, torch.int32)
        hipc('mg_plane_flags', ptr(w), c_int(P), c_int(H * W_), ptr(flags), st())
        sums = ARENA.zeros(32 * 16, dev)                                # 32 replicas x [l1, grad, w, lap0, w0, lap1, w1, lap2, w2, pad]: see mg_loss_finish
        d = torch.empty((P, H, W_), dtype=torch.float32, device=dev)
        if pvalid is not None:
            assert pvalid.dtype == torch.int32 and pvalid.numel() == P and pvalid.is_contiguous()
        hipc('mg_loss_point_fwd', ptr(p), ptr(t), ptr(w), ptr(flags), c_int(P), c_int(H), c_int(W_), ptr(d), ptr(sums), ptr(pvalid), st())
        x, h, ww = d, H, W_
        Gs = []
        for lvl in range(3):
            down = torch.empty((P, h // 2, ww // 2), dtype=torch.float32, device=dev)
            hipc('mg_pyr_down', ptr(x), ptr(flags), c_int(P), c_int(h), c_int(ww), ptr(down), st())
            G = torch.empty((P, h, ww), dtype=torch.float32, device=dev)
            hipc('mg_pyr_lap_fwd', ptr(x), ptr(down), ptr(w), c_int(lvl), c_int(H), c_int(W_), ptr(flags), c_int(P), c_int(h), c_int(ww),
                 ptr(G), ptr(sums[3 + 2 * lvl:]), st())
            Gs.append(G)
            x, h, ww = down, h // 2, ww // 2
        out = torch.empty(3, dtype=torch.float32, device=dev)                   # (rec, lap, grad)
        hipc('mg_loss_finish', ptr(sums), ptr(out), st())
        ctx.save_for_backward(p, t, w, flags, sums, *Gs)
        ctx.pvalid = pvalid
        ctx.shape = pred.shape
        return out

    @staticmethod
    def backward(ctx, g):
        p, t, w, flags, sums, G0, G1, G2 = ctx.saved_tensors
        P, H, W_ = G0.shape
        dev = p.device
        hipc, c_int = K.hip.call, K.c_int
        ptr, st = K.hip.ptr, K.hip.stream
        g = g.float().contiguous()
        coef = torch.empty(5, dtype=torch.float32, device=dev)
        hipc('mg_loss_coef', ptr(g), ptr(sums), ptr(coef), st())
        c_rec, c_grad, c0, c1, c2 = [coef[i:i + 1] for i in range(5)]
        new = lambda hh, ww: torch.empty((P, hh, ww), dtype=torch.float32, device=dev)
        r2 = new(H // 8, W_ // 8)
        hipc('mg_pyr_upT', ptr(G2), ptr(c2), None, ptr(flags), c_int(P), c_int(H // 4), c_int(W_ // 4), ptr(r2), st())
        dd2 = new(H // 4, W_ // 4)
        hipc('mg_pyr_downT', ptr(r2), ptr(G2), ptr(c2), ptr(flags), c_int(P), c_int(H // 4), c_int(W_ // 4), ptr(dd2), st())
        r1 = new(H // 4, W_ // 4)
        hipc('mg_pyr_upT', ptr(G1), ptr(c1), ptr(dd2), ptr(flags), c_int(P), c_int(H // 2), c_int(W_ // 2), ptr(r1), st())
        dd1 = new(H // 2, W_ // 2)
        hipc('mg_pyr_downT', ptr(r1), ptr(G1), ptr(c1), ptr(flags), c_int(P), c_int(H // 2), c_int(W_ // 2), ptr(dd1), st())
        r0 = new(H // 2, W_ // 2)
        hipc('mg_pyr_upT', ptr(G0), ptr(c0), ptr(dd1), ptr(flags), c_int(P), c_int(H), c_int(W_), ptr(r0), st())
        dd0 = new(H, W_)
        hipc('mg_pyr_downT', ptr(r0), ptr(G0), ptr(c0), ptr(flags), c_int(P), c_int(H), c_int(W_), ptr(dd0), st())
        A, B = new(H, W_), new(H, W_)
        dp = torch.empty((P, H, W_), dtype=torch.float32, device=dev)          # planes without weight are zeroed by the kernel
        hipc('mg_loss_point_bwd', ptr(p), ptr(t), ptr(w), ptr(flags), c_int(P), c_int(H), c_int(W_), ptr(c_rec), ptr(c_grad), ptr(dd0),
             ptr(A), ptr(B), ptr(dp), ptr(ctx.pvalid), st())
        return dp.view(ctx.shape), None, None, None


class MattingLossesMulti(torch.autograd.Function):
    """The fused loss pipeline of up to three output scales in ONE set of launches each way (mg_matting_losses_fwd / _bwd): preds and weights
    are S tensors of the same (.., H, W) shape, the target is shared. -> (S, 3) tensor of (rec, lap, grad) per scale; backward yields d/dpred."""

    @staticmethod
    def forward(ctx, target, pvalid, *pw):
        S = len(pw) // 2
        preds, weights = pw[:S], pw[S:]
        H, W_ = preds[0].shape[-2:]
        assert H % 8 == 0 and W_ % 8 == 0
        ps = [x.detach().float().contiguous() for x in preds]
        ws = [(x.expand_as(preds[0]) if x.shape != preds[0].shape else x).detach().float().contiguous() for x in weights]
        t = target.detach().float().contiguous()
        P = ps[0].numel() // (H * W_)
        dev = t.device
        K.hip.need_cuda(t, *ps, *ws)
        if pvalid is not None:
            assert pvalid.dtype == torch.int32 and pvalid.numel() == P and pvalid.is_contiguous()
        SP = S * P
        new = lambda hh, ww: torch.empty((SP, hh, ww), dtype=torch.float32, device=dev)      # noqa: E731
        flags = ARENA.acc(SP, dev, torch.int32)
        sums = ARENA.acc(S * 512, dev)
        d, downs = new(H, W_), [new(H >> 1, W_ >> 1), new(H >> 2, W_ >> 2), new(H >> 3, W_ >> 3)]
        Gs = [new(H, W_), new(H >> 1, W_ >> 1), new(H >> 2, W_ >> 2)]
        out = torch.empty((S, 3), dtype=torch.float32, device=dev)
        parr = (K.ctypes.c_void_p * S)(*[x.data_ptr() for x in ps])
        warr = (K.ctypes.c_void_p * S)(*[x.data_ptr() for x in ws])
        ptr = K.hip.ptr
        K.hip.call('mg_matting_losses_fwd', parr, ptr(t), warr, ptr(pvalid), K.c_int(S), K.c_int(P), K.c_int(H), K.c_int(W_), ptr(flags), ptr(d),
                   ptr(downs[0]), ptr(downs[1]), ptr(downs[2]), ptr(Gs[0]), ptr(Gs[1]), ptr(Gs[2]), ptr(sums), ptr(out), K.hip.stream())
        ctx.save_for_backward(t, flags, sums, *Gs, *ps, *ws)
        ctx.pvalid, ctx.S, ctx.shape = pvalid, S, preds[0].shape
        return out

    @staticmethod
    def backward(ctx, g):
        S = ctx.S
        saved = ctx.saved_tensors
        t, flags, sums, G0, G1, G2 = saved[:6]
        ps, ws = saved[6:6 + S], saved[6 + S:6 + 2 * S]
        SP, H, W_ = G0.shape
        P = SP // S
        dev = t.device
        new = lambda hh, ww: torch.empty((SP, hh, ww), dtype=torch.float32, device=dev)      # noqa: E731
        coef = torch.empty((S, 5), dtype=torch.float32, device=dev)
        r2, dd2, r1, dd1 = new(H >> 3, W_ >> 3), new(H >> 2, W_ >> 2), new(H >> 2, W_ >> 2), new(H >> 1, W_ >> 1)
        r0, dd0, A, B, dp = new(H >> 1, W_ >> 1), new(H, W_), new(H, W_), new(H, W_), new(H, W_)
        parr = (K.ctypes.c_void_p * S)(*[x.data_ptr() for x in ps])
        warr = (K.ctypes.c_void_p * S)(*[x.data_ptr() for x in ws])
        ptr = K.hip.ptr
        K.hip.call('mg_matting_losses_bwd', ptr(g.float().contiguous()), ptr(sums), parr, ptr(t), warr, ptr(ctx.pvalid), ptr(flags), K.c_int(S), K.c_int(P),
                   K.c_int(H), K.c_int(W_), ptr(G0), ptr(G1), ptr(G2), ptr(coef), ptr(r2), ptr(dd2), ptr(r1), ptr(dd1), ptr(r0), ptr(dd0), ptr(A), ptr(B),
                   ptr(dp), K.hip.stream())
        dps = tuple(dp[i * P:(i + 1) * P].view(ctx.shape) for i in range(S))
        return (None, None) + dps + (None,) * S


def matting_losses_multi(preds, target, weights, pvalid=None):
    """[(rec, lap, grad)] per prediction: the scales' loss pipelines as one batched set of launches."""
    if not LOSS_MULTI or len(preds) == 1 or len(preds) > 3:
        return [matting_losses(p_, target, w_, pvalid) for p_, w_ in zip(preds, weights)]
    out = MattingLossesMulti.apply(target, pvalid, *preds, *weights)
    cells = SplitGrid.apply(out)
    n = out.shape[1]
    return [cells[i * n:(i + 1) * n] for i in range(out.shape[0])]


class SplitGrid(torch.autograd.Function):
    """A small 2-D tensor -> its elements as zero-dim views, row-major. Backward: when the elements' gradients are consecutive words of ONE buffer in
    that order (the weighted sum of the loss terms hands them out that way: ScalarLinComb.backward, arch/maggie.py compute_loss orders its terms to
    match) the buffer's window IS the gradient -- no launch; otherwise they are stacked (unbind + unbind cost four stack launches per step)."""

    @staticmethod
    def forward(ctx, t):
        ctx.set_materialize_grads(False)
        ctx.shape = t.shape
        return tuple(t[i, j] for i in range(t.shape[0]) for j in range(t.shape[1]))

    @staticmethod
    def backward(ctx, *gs):
        R, C = ctx.shape
        g0 = gs[0]
        if g0 is not None and all(g is not None and g.dtype == torch.float32 and g.numel() == 1 and g.is_cuda and
                                  g.untyped_storage().data_ptr() == g0.untyped_storage().data_ptr() and g.data_ptr() == g0.data_ptr() + 4 * i
                                  for i, g in enumerate(gs)):
            return g0.as_strided((R, C), (C, 1))
        live = next((g for g in gs if g is not None), None)
        if live is None:
            return None
        return torch.stack([(g if g is not None else torch.zeros_like(live)).reshape(()).to(live.dtype) for g in gs]).view(R, C)


LOSS_MULTI = os.environ.get('MAGGIE_LOSS_MULTI', '1') != '0'


def os8_weight(alphas, a8, reweight=True, pvalid=None):
    """Loss weight of the OS8 prediction (arch/maggie.py:271-281): [plane has ground truth] + [pixel in the unknown band of gt or a8]."""
    gt, a = alphas.detach().contiguous(), a8.detach().contiguous()
    H, W_ = gt.shape[-2:]
    P = gt.numel() // (H * W_)
    out = torch.empty_like(a)
    flags = torch.empty(P, dtype=torch.int32, device=gt.device)
    K.hip.call('mg_os8_weight_ex', K.hip.ptr(gt), K.hip.ptr(a), K.c_int(P), K.c_long(H * W_), K.c_int(int(bool(reweight))), K.hip.ptr(flags),
               K.hip.ptr(out), K.hip.ptr(pvalid), K.hip.stream())
    return out


def matting_losses(pred, target, weight, pvalid=None):
    """-> (rec, lap, grad) scalars. `pvalid` (int32 [planes]): planes with 0 are evaluated as if pred were zero there (`pred * valid_masks`)."""
    out = MattingLosses.apply(pred, target, weight.expand_as(pred) if weight.shape != pred.shape else weight, pvalid)
    return out.unbind(0)


# ----------------------------------------------------------------------------------------------------------------------
# instance-token <-> feature cross attention (maggie_amd/csrc/attention.hip)
# ----------------------------------------------------------------------------------------------------------------------

class AttnTokensFromFeatures(torch.autograd.Function):
    """p = softmax_l((qk . F^T + btab[:, ids]) * scale), ctx = p F.   qk (B,T,D), btab (B,T,NID), feat (B,L,D), ids (B,L) int32."""

    @staticmethod
    def forward(ctx, qk, btab, feat, ids, scale):
        qk, btab, feat = qk.float().contiguous(), btab.float().contiguous(), feat.float().contiguous()
        p, c = K.attn_tok_fwd(qk, btab, feat, ids, scale)
        ctx.save_for_backward(qk, feat, ids, p)
        ctx.scale, ctx.nid = scale, btab.shape[2]
        return p, c

    @staticmethod
    def backward(ctx, dp, dctx):
        qk, feat, ids, p = ctx.saved_tensors
        dctx = torch.zeros_like(qk) if dctx is None else dctx.float().contiguous()
        dp = None if dp is None else dp.float().contiguous()
        dqk, dbtab, dfeat = K.attn_tok_bwd(p, feat, qk, ids, dctx, dp, ctx.scale, ctx.nid)
        return dqk, dbtab, dfeat, None, None


class AttnFeaturesFromTokens(torch.autograd.Function):
    """out = softmax_t((F . kq^T + b2[ids]) * scale, masked) vp + obias.   kq / vp (B,T,D), b2 (B,NID,T) -- or (B,T,NID) with tn=True, the layout
    the token-side linear writes the table in (no transposed copy either way) --, pad (B,T) bool or None."""

    @staticmethod
    def forward(ctx, feat, kq, b2, vp, obias, pad, ids, scale, tn=False):
        feat, kq, b2, vp = feat.float().contiguous(), kq.float().contiguous(), b2.float().contiguous(), vp.float().contiguous()
        ob = None if obias is None else obias.float().contiguous()
        pd = None if pad is None else as_u8(pad)
        out, p = K.attn_feat_fwd(feat, kq, b2, vp, ob, pd, ids, scale, tn)
        ctx.save_for_backward(feat, kq, vp, ids, p)
        ctx.scale, ctx.nid, ctx.has_bias, ctx.tn = scale, (b2.shape[2] if tn else b2.shape[1]), obias is not None, bool(tn)
        return out

    @staticmethod
    def backward(ctx, dout):
        feat, kq, vp, ids, p = ctx.saved_tensors
        dfeat, dkq, dvp, db2, dob = K.attn_feat_bwd(dout.float().contiguous(), p, feat, kq, vp, ids, ctx.scale, ctx.nid, ctx.has_bias, ctx.tn)
        return dfeat, dkq, db2, dvp, dob, None, None, None, None


def attn_tokens_from_features(qk, btab, feat, ids, scale):
    return AttnTokensFromFeatures.apply(qk, btab, feat, ids, scale)


def attn_features_from_tokens(feat, kq, b2, vp, obias, pad, ids, scale, tn=False):
    return AttnFeaturesFromTokens.apply(feat, kq, b2, vp, obias, pad, ids, scale, tn)


# ----------------------------------------------------------------------------------------------------------------------
# ConvGRU gate math around the two gate convolutions (maggie_amd/csrc/temporal.hip)
# ----------------------------------------------------------------------------------------------------------------------

class GruGate(torch.autograd.Function):
    """[x | sigmoid(r) * h] from the first gate conv's pre-activation rz = [r | z] (.., 2C)."""

    @staticmethod
    def forward(ctx, rz, x, h):
        rz, x, h = rz.contiguous(), x.contiguous(), h.contiguous()
        ctx.save_for_backward(rz, h)
        return K.gru_gate_fwd(rz, x, h)

    @staticmethod
    def backward(ctx, dxrh):
        rz, h = ctx.saved_tensors
        return K.gru_gate_bwd(dxrh.contiguous(), rz, h)


class GruOut(torch.autograd.Function):
    """(1 - sigmoid(z)) * h + sigmoid(z) * tanh(cpre)."""

    @staticmethod
    def forward(ctx, rz, cpre, h):
        rz, cpre, h = rz.contiguous(), cpre.contiguous(), h.contiguous()
        ctx.save_for_backward(rz, cpre, h)
        return K.gru_out_fwd(rz, cpre, h)

    @staticmethod
    def backward(ctx, dhn):
        rz, cpre, h = ctx.saved_tensors
        return K.gru_out_bwd(dhn.contiguous(), rz, cpre, h)


# ----------------------------------------------------------------------------------------------------------------------
# temporal-consistency tail of the video model (maggie_amd/csrc/temporal2.hip)
# ----------------------------------------------------------------------------------------------------------------------

class BiFuse(torch.autograd.Function):
    """bidirectional_fusion (resnet_inst_matt_spconv_temp.py:35-79) given the 2(T-1) difference logits: preds (B,T,NI,H,W) fp32,
    diffs (2(T-1), B, 1, H, W) fp32 -> fused (B,T,NI,H,W), forward / backward difference logits (B,T,1,H,W) and their sigmoids."""

    @staticmethod
    def forward(ctx, preds, diffs):
        preds, diffs = preds.float().contiguous(), diffs.float().contiguous()
        B, T, NI, H, W = preds.shape
        HW = H * W
        fused = torch.empty_like(preds)
        aux = torch.empty((4, B, T, 1, H, W), dtype=torch.float32, device=preds.device)
        K.hip.call('mg_bifuse_fwd', K.hip.ptr(preds), K.hip.ptr(diffs), K.c_int(B), K.c_int(T), K.c_int(NI), K.c_long(HW), K.hip.ptr(fused),
                   K.hip.ptr(aux[0]), K.hip.ptr(aux[1]), K.hip.ptr(aux[2]), K.hip.ptr(aux[3]), K.hip.stream())
        ctx.save_for_backward(preds, diffs)
        ctx.mark_non_differentiable(aux)
        return fused, aux

    @staticmethod
    def backward(ctx, dfused, _daux):
        preds, diffs = ctx.saved_tensors
        B, T, NI, H, W = preds.shape
        dp, dd = torch.empty_like(preds), torch.empty_like(diffs)
        K.hip.call('mg_bifuse_bwd', K.hip.ptr(dfused.float().contiguous()), K.hip.ptr(preds), K.hip.ptr(diffs), K.c_int(B), K.c_int(T), K.c_int(NI),
                   K.c_long(H * W), K.hip.ptr(dp), K.hip.ptr(dd), K.hip.stream())
        return dp, dd


def _frames(t):
    """(B, T, ...) fp32 tensor whose frames are contiguous blocks -> (tensor, batch stride, T, elements per frame); a slice along T
    (x[:, 1:], x[:, :-1]) is used in place: only the batch stride differs."""
    if t.dtype != torch.float32:
        t = t.float()
    E = 1
    for d in t.shape[2:]:
        E *= d
    inner_ok = t[0, 0].is_contiguous() and (t.shape[1] == 1 or t.stride(1) == E)
    if not inner_ok:
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else t.shape[1] * E), t.shape[1], E


class DtSSD(torch.autograd.Function):
    """loss_dtSSD (maggie/network/loss.py:7-16); sig: pred = sigmoid(logits) (loss_temporal_sparsity); mask None = ones."""

    @staticmethod
    def forward(ctx, pred, gt, mask, sig):
        p, pbs, T, E = _frames(pred)
        g, gbs, _, _ = _frames(gt.detach())
        m, mbs = (None, 0) if mask is None else _frames(mask.detach())[:2]
        B = p.shape[0]
        sums = torch.empty(2, dtype=torch.float32, device=p.device)
        K.hip.call('mg_dtssd_fwd', K.hip.ptr(p), K.c_long(pbs), K.hip.ptr(g), K.c_long(gbs), K.hip.ptr(m), K.c_long(mbs), K.c_int(B), K.c_int(T),
                   K.c_long(E), K.c_int(int(sig)), K.hip.ptr(sums), K.hip.stream())
        ctx.save_for_backward(p, g, m, sums)
        ctx.meta = (pbs, gbs, mbs, B, T, E, int(sig), pred.shape)
        return sums[0] / sums[1]

    @staticmethod
    def backward(ctx, gout):
        p, g, m, sums = ctx.saved_tensors
        pbs, gbs, mbs, B, T, E, sig, shape = ctx.meta
        dp = torch.empty(shape, dtype=torch.float32, device=p.device)
        K.hip.call('mg_dtssd_bwd', K.hip.ptr(p), K.c_long(pbs), K.hip.ptr(g), K.c_long(gbs), K.hip.ptr(m), K.c_long(mbs), K.c_int(B), K.c_int(T),
                   K.c_long(E), K.c_int(sig), K.hip.ptr(sums), K.hip.ptr(gout.float().contiguous().view(1)), K.hip.ptr(dp), K.c_long(T * E), K.hip.stream())
        return dp, None, None, None


def dtssd_loss(pred, gt, mask=None, sig=False):
    return DtSSD.apply(pred, gt, mask, sig)


class AttenGuidanceLoss(torch.autograd.Function):
    """compute_atten_loss (maggie/network/module/instance_matte_decoder.py): scale * sum((sum_l gm != 0) - sum_l gm * att) over the (b, slot) rows;
    two HIP launches forward, one backward (mg_atten_loss_*)."""

    @staticmethod
    def forward(ctx, gm, att, scale):
        gm2 = gm.detach().float().contiguous()
        a2 = att.float().contiguous()
        L = gm2.shape[-1]
        rows = gm2.numel() // L
        out = torch.empty(1, dtype=torch.float32, device=a2.device)
        terms = torch.empty(rows, dtype=torch.float32, device=a2.device)
        K.hip.need_cuda(a2)
        K.hip.call('mg_atten_loss_fwd', K.hip.ptr(gm2), K.hip.ptr(a2), K.c_int(rows), K.c_long(L), K.c_float(float(scale)), K.hip.ptr(terms),
                   K.hip.ptr(out), K.hip.stream())
        ctx.save_for_backward(gm2)
        ctx.meta = (float(scale), att.shape, att.dtype)
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        gm2, = ctx.saved_tensors
        scale, shape, dtype = ctx.meta
        datt = torch.empty(shape, dtype=torch.float32, device=gm2.device)
        K.hip.call('mg_atten_loss_bwd', K.hip.ptr(gm2), K.hip.ptr(gout.float().contiguous().view(1)), K.c_float(scale), K.c_long(gm2.numel()),
                   K.hip.ptr(datt), K.hip.stream())
        return None, datt if dtype == torch.float32 else datt.to(dtype), None


def atten_guidance_loss(gm, att, scale):
    return AttenGuidanceLoss.apply(gm, att, scale)


class ScalarLinComb(torch.autograd.Function):
    """sum_i coef[i] * t_i over up to 16 one-element CUDA tensors in ONE launch (and one for the backward): the loss sums of arch/maggie.py:283-300."""

    @staticmethod
    def forward(ctx, coefs, *terms):
        n = len(terms)
        ts = [t.detach().float().reshape(1) for t in terms]
        out = torch.empty(1, dtype=torch.float32, device=ts[0].device)
        K.hip.need_cuda(ts[0])
        ptrs = (K.ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        cf = (K.ctypes.c_float * n)(*[float(c) for c in coefs])
        K.hip.call('mg_scalar_lincomb', ptrs, cf, K.c_int(n), K.hip.ptr(out), K.hip.stream())
        ctx.coefs = tuple(float(c) for c in coefs)
        ctx.shapes = [t.shape for t in terms]
        ctx.keep = ts                                   # the launch read them through raw pointers
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        n = len(ctx.coefs)
        gin = torch.empty(n, dtype=torch.float32, device=gout.device)
        cf = (K.ctypes.c_float * n)(*ctx.coefs)
        K.hip.call('mg_scalar_lincomb_bwd', cf, K.c_int(n), K.hip.ptr(gout.float().contiguous().view(1)), K.hip.ptr(gin), K.hip.stream())
        return (None,) + tuple(g.view(sh) for g, sh in zip(gin.unbind(0), ctx.shapes))


def scalar_lincomb(terms, coefs):
    """Weighted sum of one-element tensors; python numbers among `terms` are folded on the host."""
    const = sum(float(t) * float(c) for t, c in zip(terms, coefs) if not torch.is_tensor(t))
    tt = [(t, c) for t, c in zip(terms, coefs) if torch.is_tensor(t)]
    if not tt:
        return const
    out = ScalarLinComb.apply([c for _, c in tt], *[t for t, _ in tt])
    return out + const if const else out


class BCELogitsMean(torch.autograd.Function):
    """F.binary_cross_entropy_with_logits(x, y, reduction='mean') over (B, T, ...) frame slices without copies."""

    @staticmethod
    def forward(ctx, x, y):
        xx, xbs, T, E = _frames(x)
        yy, ybs, _, _ = _frames(y.detach())
        B = xx.shape[0]
        s = torch.empty(1, dtype=torch.float32, device=xx.device)
        K.hip.call('mg_bce_logits_fwd', K.hip.ptr(xx), K.c_long(xbs), K.hip.ptr(yy), K.c_long(ybs), K.c_int(B), K.c_long(T * E), K.hip.ptr(s), K.hip.stream())
        ctx.save_for_backward(xx, yy)
        ctx.meta = (xbs, ybs, B, T * E, x.shape)
        return s[0] / float(B * T * E)

    @staticmethod
    def backward(ctx, gout):
        xx, yy = ctx.saved_tensors
        xbs, ybs, B, TE, shape = ctx.meta
        dx = torch.empty(shape, dtype=torch.float32, device=xx.device)
        K.hip.call('mg_bce_logits_bwd', K.hip.ptr(xx), K.c_long(xbs), K.hip.ptr(yy), K.c_long(ybs), K.c_int(B), K.c_long(TE),
                   K.hip.ptr(gout.float().contiguous().view(1)), K.hip.ptr(dx), K.c_long(TE), K.hip.stream())
        return dx, None


def bce_logits_mean(x, y):
    return BCELogitsMean.apply(x, y)


def temporal_crop_(alpha, bits, sigma=3, thr=0.1, pad=30):
    """Eval-time bounding-box crop of the video decoder (resnet_inst_matt_spconv_temp.py:115-142): in place on the coarse alpha planes
    (N, n_i, H, W) fp32 and on the detail bit planes (N*n_i, H, Ww)."""
    a = alpha.contiguous()
    assert a.dtype == torch.float32 and a.data_ptr() == alpha.data_ptr()
    H, W_ = a.shape[-2:]
    P = a.numel() // (H * W_)
    scratch = torch.empty_like(a)
    box = torch.empty((P, 4), dtype=torch.int32, device=a.device)
    K.hip.call('mg_temporal_crop', K.hip.ptr(a), K.hip.ptr(bits), K.c_int(P), K.c_int(H), K.c_int(W_), K.c_float(float(sigma)), K.c_float(float(thr)),
               K.c_int(int(pad)), K.hip.ptr(scratch), K.hip.ptr(box), K.hip.stream())
    return alpha, bits


# ----------------------------------------------------------------------------------------------------------------------
# token side of the instance matte decoder (maggie_amd/csrc/token_side.hip)
# ----------------------------------------------------------------------------------------------------------------------

class GradSlots:
    """One buffer per backward pass that will hold the gradients of the k slices of a PACKED parameter (nn.MultiheadAttention's in_proj_weight /
    in_proj_bias). The kernels that produce a slice's gradient write it into the buffer's slice (grad_slot_out) and return that very view;
    SplitPacked.backward then hands the buffer back as the packed parameter's gradient -- no `stack` of three separate tensors (18 copy launches
    per training step over the nine attention layers of the instance matte decoder)."""
    __slots__ = ('k', 'buf', 'taken')

    def __init__(self, k):
        self.k, self.buf, self.taken = k, None, set()


def grad_slot_out(slot, shape, device):
    """-> the fp32 destination inside the packed buffer for a gradient of `shape`, or None (no slot / already handed out in this pass / another shape):
    the caller then allocates its own tensor and SplitPacked.backward falls back to stacking."""
    if slot is None:
        return None
    holder, i = slot
    shape = tuple(shape)
    if i in holder.taken:
        return None
    if holder.buf is None:
        holder.buf = torch.empty((holder.k,) + shape, dtype=torch.float32, device=device)
    elif tuple(holder.buf.shape[1:]) != shape or holder.buf.device != device:
        return None
    holder.taken.add(i)
    return holder.buf[i]


class SplitPacked(torch.autograd.Function):
    """P (k * d, ...) -> its k slices (views). Backward: the slices' gradients, already sitting in the slices of ONE buffer when their producers took
    the offered slots (GradSlots), else stacked."""

    @staticmethod
    def forward(ctx, P, k, holder):
        ctx.set_materialize_grads(False)
        ctx.holder, ctx.k, ctx.shape = holder, k, P.shape
        v = P.view(k, P.shape[0] // k, *P.shape[1:])
        return tuple(v[i] for i in range(k))

    @staticmethod
    def backward(ctx, *gs):
        holder = ctx.holder
        buf, holder.buf = holder.buf, None
        holder.taken.clear()
        if buf is not None and all(g is not None and g.dtype == buf.dtype and g.shape == buf.shape[1:] and g.data_ptr() == buf[i].data_ptr()
                                   for i, g in enumerate(gs)):
            return buf.view(ctx.shape), None, None
        live = [g for g in gs if g is not None]
        if not live:
            return None, None, None
        return torch.stack([g if g is not None else torch.zeros_like(live[0]) for g in gs]).view(ctx.shape), None, None


GRAD_SLOTS = os.environ.get('MAGGIE_GRAD_SLOTS', '1') != '0'      # 0: unbind + stack (A/B switch)


def split_packed(P, k):
    """The k equal slices of a packed parameter along dim 0, each carrying the slot its gradient should be written to (`_mg_gslot`)."""
    if not (GRAD_SLOTS and P.requires_grad and torch.is_grad_enabled() and P.is_cuda and P.dtype == torch.float32):
        return P.view(k, P.shape[0] // k, *P.shape[1:]).unbind(0)
    holder = GradSlots(k)
    outs = SplitPacked.apply(P, k, holder)
    for i, o in enumerate(outs):
        o._mg_gslot = (holder, i)
    return outs


def as_u8(mask):
    """A 0/1 mask as contiguous uint8: a bool tensor is re-interpreted in place (one byte per element, 0 / 1) -- no cast kernel per consumer (the token
    padding mask met six of them per step)."""
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)
    return mask.to(torch.uint8).contiguous()


class TokenLinear(torch.autograd.Function):
    """y = LN( res + act( (x + xadd) W^T + b ) ) over (..., K) -> (..., N) fp32; every optional piece may be None. One HIP launch each way
    (the reference: up to 2 adds + cuBLAS + bias + ReLU + add + LayerNorm forward, twice that backward)."""

    @staticmethod
    def forward(ctx, x, xadd, W, b, res, relu, gamma, beta, eps, wt=False):
        if relu and (res is not None or gamma is not None):
            raise K.hip.MaggieHipError('TokenLinear: ReLU is only fused for a plain linear layer (no residual / LayerNorm behind it)')
        shape = x.shape
        Kd, N = shape[-1], (W.shape[1] if wt else W.shape[0])                 # wt: W is (K, N) and y = x W
        f = lambda t: None if t is None else t.detach().float().contiguous()       # noqa: E731
        x2, xa, W_, b_, r_, g_, be_ = f(x).view(-1, Kd), f(xadd), f(W), f(b), f(res), f(gamma), f(beta)
        if xa is not None:
            xa = xa.expand(shape).contiguous().view(-1, Kd) if xa.shape != shape else xa.view(-1, Kd)
        if r_ is not None:
            r_ = r_.view(-1, N)
        R = x2.shape[0]
        y = torch.empty((R, N), dtype=torch.float32, device=x.device)
        z = torch.empty((R, N), dtype=torch.float32, device=x.device) if g_ is not None else None
        rstat = torch.empty((R, 2), dtype=torch.float32, device=x.device) if g_ is not None else None
        K.hip.call('mg_token_linear_fwd_ex', K.hip.ptr(x2), K.hip.ptr(xa), K.hip.ptr(W_), K.hip.ptr(b_), K.hip.ptr(r_), K.c_int(int(bool(relu))),
                   K.hip.ptr(g_), K.hip.ptr(be_), K.c_float(float(eps)), K.hip.ptr(y), K.hip.ptr(z), K.hip.ptr(rstat), K.c_int(R), K.c_int(Kd),
                   K.c_int(N), K.c_int(int(bool(wt))), K.hip.stream())
        ctx.save_for_backward(x2, xa, W_, y if relu else None, g_, z, rstat)
        ctx.wt = bool(wt)
        ctx.gslots = (getattr(W, '_mg_gslot', None), getattr(b, '_mg_gslot', None))      # slices of a packed parameter: gradients go into its buffer
        ctx.meta = (shape, R, Kd, N, bool(relu), b is not None, res is not None, None if xadd is None else xadd.shape)
        return y.view(*shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, xa, W_, yout, g_, z, rstat = ctx.saved_tensors
        shape, R, Kd, N, relu, has_b, has_res, xadd_shape = ctx.meta
        dev = dy.device
        dy2 = dy.float().contiguous().view(R, N)
        need_dx = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        dx = torch.empty((R, Kd), dtype=torch.float32, device=dev) if need_dx else None
        dW = grad_slot_out(ctx.gslots[0], (Kd, N) if ctx.wt else (N, Kd), dev)
        if dW is None:
            dW = torch.empty((Kd, N) if ctx.wt else (N, Kd), dtype=torch.float32, device=dev)
        db = None
        if has_b:
            db = grad_slot_out(ctx.gslots[1], (N,), dev)
            if db is None:
                db = torch.empty(N, dtype=torch.float32, device=dev)
        plain = g_ is None and not relu                      # dz == dy: no dz kernel, the residual gradient IS dy
        dres = torch.empty((R, N), dtype=torch.float32, device=dev) if (has_res and not plain) else None
        dgb = torch.empty(2 * N, dtype=torch.float32, device=dev) if g_ is not None else None
        dz = dy2 if plain else torch.empty((R, N), dtype=torch.float32, device=dev)
        K.hip.call('mg_token_linear_bwd_ex', K.hip.ptr(dy2), K.hip.ptr(x2), K.hip.ptr(xa), K.hip.ptr(W_), K.hip.ptr(yout), K.c_int(int(relu)), K.hip.ptr(g_),
                   K.hip.ptr(z), K.hip.ptr(rstat), K.hip.ptr(dx), K.hip.ptr(dW), K.hip.ptr(db), K.hip.ptr(dres),
                   K.hip.ptr(None if dgb is None else dgb[:N]), K.hip.ptr(None if dgb is None else dgb[N:]), K.hip.ptr(dz), K.c_int(R), K.c_int(Kd),
                   K.c_int(N), K.c_int(int(ctx.wt)), K.hip.stream())
        dxv = None if dx is None else dx.view(shape)
        dxadd = None
        if xadd_shape is not None and ctx.needs_input_grad[1]:
            dxadd = dxv if tuple(xadd_shape) == tuple(shape) else dxv.sum_to_size(xadd_shape)
        if plain and has_res:
            dres = dy2
        return (dxv if ctx.needs_input_grad[0] else None, dxadd, dW, db, None if dres is None else dres.view(*shape[:-1], N), None,
                None if dgb is None else dgb[:N], None if dgb is None else dgb[N:], None, None)


def token_linear(x, W, b=None, xadd=None, res=None, relu=False, ln=None, wt=False):
    """`ln`: an nn.LayerNorm (weight, bias, eps) applied to (res + linear output). `wt`: W is (K, N) and the product is x W (no transposed copy)."""
    g, be, eps = (ln.weight, ln.bias, ln.eps) if ln is not None else (None, None, 0.0)
    return TokenLinear.apply(x, xadd, W, b, res, relu, g, be, eps, wt)


FAN_OUT = os.environ.get('MAGGIE_FAN_OUT', '1') != '0'


class SpatialMean(torch.autograd.Function):
    """AdaptiveAvgPool2d(1) of an NHWC map: (N, H, W, C) -> (N, 1, 1, C) in the map's dtype, fp32 sums in a fixed order; one launch each way."""

    @staticmethod
    def forward(ctx, x):
        N, H, W_, C = x.shape
        ctx.hw = (H, W_)
        return K.spatial_mean(x.view(N, H * W_, C), N, H * W_).view(N, 1, 1, C)

    @staticmethod
    def backward(ctx, dy):
        H, W_ = ctx.hw
        N, C = dy.shape[0], dy.shape[-1]
        return K.spatial_mean(dy.reshape(N, C), N, H * W_, mode=1).view(N, H, W_, C)


def spatial_mean(x):
    return SpatialMean.apply(x)


class SpatialBroadcast(torch.autograd.Function):
    """(N, 1, 1, C) -> its expansion over an (H, W) map (a view: the nearest up-sampling of a 1 x 1 map). Backward: the sum over the map in ONE launch,
    read in place from a channel slice of a wider gradient (torch: a strided reduction, 18 us for the ASPP's pooled branch)."""

    @staticmethod
    def forward(ctx, x, H, W_):
        return x.expand(x.shape[0], H, W_, x.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        N, H, W_, C = dy.shape
        rows = dy.as_strided((N, H * W_, C), (dy.stride(0), dy.stride(2), dy.stride(3))) if dy.stride(1) == W_ * dy.stride(2) else dy.reshape(N, H * W_, C)
        return K.spatial_mean(rows, N, H * W_, mode=2).view(N, 1, 1, C), None, None


def spatial_broadcast(x, H, W_):
    return SpatialBroadcast.apply(x, H, W_)


class FanOut(torch.autograd.Function):
    """k aliases of one tensor, one per consumer: the k gradients then arrive HERE together and are added by one launch in alias order
    (mg_sum_k), instead of the k - 1 pairwise add kernels the autograd engine issues for a tensor it sees consumed k times. The token side of the
    instance matte decoder uses its tokens, their position embedding and the ID table 4-13 times each (mask_attention.py:63-133)."""

    @staticmethod
    def forward(ctx, t, k):
        ctx.set_materialize_grads(False)
        ctx.gslot = getattr(t, '_mg_gslot', None)                 # a slice of a packed parameter: the sum is written into the packed buffer's slice
        return tuple(t.view_as(t) for _ in range(k))

    @staticmethod
    def backward(ctx, *gs):
        live = [g for g in gs if g is not None]
        if not live:
            return None, None
        if ctx.gslot is not None and len(live) > 1 and all(g.is_cuda and g.dtype == torch.float32 and g.shape == live[0].shape for g in live) \
                and len(live) <= 16:
            out = grad_slot_out(ctx.gslot, live[0].shape, live[0].device)
            if out is not None:
                return K.sum_k([g.contiguous() for g in live], out=out), None
        ok = all(g.is_cuda and g.dtype == live[0].dtype and g.shape == live[0].shape for g in live) and \
            live[0].dtype in (torch.float32, torch.bfloat16, torch.float16)
        if not ok:
            total = live[0]
            for g in live[1:]:
                total = total + g
            return total, None
        live = [g.contiguous() for g in live]
        while len(live) > 1:                                      # (more than 16 consumers: the first 16 collapse into one term)
            live = [K.sum_k(live[:16])] + live[16:]
        return live[0], None


class Fan:
    """Hands out the aliases of FanOut one by one: `f = Fan(t, k)`, then `f()` wherever `t` would have been passed to a consumer. A tensor that needs
    no gradient (or MAGGIE_FAN_OUT=0) is handed out as it is; running out of aliases falls back to the tensor itself (autograd then adds that
    consumer's gradient the ordinary way)."""
    __slots__ = ('t', 'outs')

    def __init__(self, t, k):
        self.t = t
        use = FAN_OUT and t is not None and torch.is_grad_enabled() and t.requires_grad and t.is_cuda and \
            t.dtype in (torch.float32, torch.bfloat16, torch.float16) and \
            (k > 2 or (k == 2 and getattr(t, '_mg_gslot', None) is not None))
        self.outs = list(FanOut.apply(t, k)) if use else None

    def __call__(self):
        if self.outs:
            return self.outs.pop()
        return self.t


def take(x):
    """x() for a Fan, x for a tensor / None: layer code that accepts either."""
    return x() if isinstance(x, Fan) else x


class TokenLinearMulti(torch.autograd.Function):
    """Up to six INDEPENDENT TokenLinear layers in one launch each way (mg_token_linear_multi_fwd / _bwd): inputs are 7 slots per layer
    (x, xadd, W, b, res, gamma, beta; None where absent), `specs` = per layer (relu, eps, wt)."""

    @staticmethod
    def forward(ctx, specs, *slots):
        n = len(specs)
        f = lambda t: None if t is None else t.detach().float().contiguous()       # noqa: E731
        ops = (K.hip.TokLin * n)()
        saved, metas, outs = [], [], []
        for i, (relu, eps, wt) in enumerate(specs):
            x, xadd, W, b, res, gamma, beta = slots[7 * i:7 * i + 7]
            if relu and (res is not None or gamma is not None):
                raise K.hip.MaggieHipError('TokenLinear: ReLU is only fused for a plain linear layer (no residual / LayerNorm behind it)')
            shape = x.shape
            Kd, N = shape[-1], (W.shape[1] if wt else W.shape[0])
            x2, xa, W_, b_, r_, g_, be_ = f(x).view(-1, Kd), f(xadd), f(W), f(b), f(res), f(gamma), f(beta)
            if xa is not None:
                xa = xa.expand(shape).contiguous().view(-1, Kd) if xa.shape != shape else xa.view(-1, Kd)
            if r_ is not None:
                r_ = r_.view(-1, N)
            R = x2.shape[0]
            y = torch.empty((R, N), dtype=torch.float32, device=x.device)
            z = torch.empty((R, N), dtype=torch.float32, device=x.device) if g_ is not None else None
            rstat = torch.empty((R, 2), dtype=torch.float32, device=x.device) if g_ is not None else None
            o = ops[i]
            o.x, o.xadd, o.W, o.bias, o.res, o.gamma, o.beta = (K.hip.ptr(t) for t in (x2, xa, W_, b_, r_, g_, be_))
            o.y, o.z, o.rstat = K.hip.ptr(y), K.hip.ptr(z), K.hip.ptr(rstat)
            o.R, o.K, o.N, o.relu, o.wt, o.eps = R, Kd, N, int(bool(relu)), int(bool(wt)), float(eps)
            saved += [x2, xa, W_, y if relu else None, g_, z, rstat]
            metas.append((shape, R, Kd, N, bool(relu), b is not None, res is not None, None if xadd is None else xadd.shape, bool(wt),
                          getattr(W, '_mg_gslot', None), getattr(b, '_mg_gslot', None)))
            outs.append(y.view(*shape[:-1], N))
        K.hip.need_cuda(*[t for t in saved if t is not None])
        K.hip.call('mg_token_linear_multi_fwd', ops, K.c_int(n), K.hip.stream())
        ctx.save_for_backward(*saved)
        ctx.metas = metas
        # layers reading the SAME tensor (no xadd on either): the backward forms ONE input gradient for it (dx_pair) instead of two and an autograd add
        ctx.pairs = {}
        if TOKEN_DX_PAIRS:
            taken = set()
            for i in range(n):
                for j in range(i + 1, n):
                    xi, xj = slots[7 * i], slots[7 * j]
                    if i not in taken and j not in taken and xi is xj and slots[7 * i + 1] is None and slots[7 * j + 1] is None \
                            and ctx.needs_input_grad[1 + 7 * i] and ctx.needs_input_grad[1 + 7 * j]:
                        ctx.pairs[i] = j
                        taken.update((i, j))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        saved, metas = ctx.saved_tensors, ctx.metas
        n = len(metas)
        ops = (K.hip.TokLin * n)()
        keep, results = [], []
        for i, (shape, R, Kd, N, relu, has_b, has_res, xadd_shape, wt, slot_w, slot_b) in enumerate(metas):
            x2, xa, W_, yout, g_, z, rstat = saved[7 * i:7 * i + 7]
            dev = x2.device
            dy = dys[i]
            dy2 = (torch.zeros((R, N), dtype=torch.float32, device=dev) if dy is None else dy.float().contiguous().view(R, N))
            need_dx = (ctx.needs_input_grad[1 + 7 * i] or ctx.needs_input_grad[2 + 7 * i]) and i not in ctx.pairs.values()
            dx = torch.empty((R, Kd), dtype=torch.float32, device=dev) if need_dx else None
            dW = grad_slot_out(slot_w, (Kd, N) if wt else (N, Kd), dev)
            if dW is None:
                dW = torch.empty((Kd, N) if wt else (N, Kd), dtype=torch.float32, device=dev)
            db = None
            if has_b:
                db = grad_slot_out(slot_b, (N,), dev)
                if db is None:
                    db = torch.empty(N, dtype=torch.float32, device=dev)
            plain = g_ is None and not relu
            dres = torch.empty((R, N), dtype=torch.float32, device=dev) if (has_res and not plain) else None
            dgb = torch.empty(2 * N, dtype=torch.float32, device=dev) if g_ is not None else None
            dz = dy2 if plain else torch.empty((R, N), dtype=torch.float32, device=dev)
            o = ops[i]
            o.x, o.xadd, o.W, o.gamma, o.z, o.rstat = (K.hip.ptr(t) for t in (x2, xa, W_, g_, z, rstat))
            o.dy, o.yout, o.dx, o.dW, o.db, o.dres, o.dz = (K.hip.ptr(t) for t in (dy2, yout, dx, dW, db, dres, dz))
            o.dgamma, o.dbeta = K.hip.ptr(None if dgb is None else dgb[:N]), K.hip.ptr(None if dgb is None else dgb[N:])
            o.R, o.K, o.N, o.relu, o.wt = R, Kd, N, int(relu), int(wt)
            o.dx_pair = ctx.pairs[i] + 1 if i in ctx.pairs else 0
            keep += [dy2, dx, dW, db, dres, dgb, dz]
            dxv = None if dx is None else dx.view(shape)
            dxadd = None
            if xadd_shape is not None and ctx.needs_input_grad[2 + 7 * i]:
                dxadd = dxv if tuple(xadd_shape) == tuple(shape) else dxv.sum_to_size(xadd_shape)
            if plain and has_res:
                dres = dy2
            results += [dxv if (ctx.needs_input_grad[1 + 7 * i] and dxv is not None) else None, dxadd, dW, db, None if dres is None else dres.view(*shape[:-1], N),
                        None if dgb is None else dgb[:N], None if dgb is None else dgb[N:]]
        K.hip.call('mg_token_linear_multi_bwd', ops, K.c_int(n), K.hip.stream())
        return (None,) + tuple(results)


TOKEN_MULTI = os.environ.get('MAGGIE_TOKEN_MULTI', '1') != '0'
TOKEN_DX_PAIRS = os.environ.get('MAGGIE_TOKEN_DX_PAIRS', '1') != '0'     # 0: one input gradient per layer, autograd adds those of a shared tensor (A/B)


TOKEN_MULTI_MAX = 6            # csrc/token_side.hip: TOK_MULTI
TOKEN_XBLOCK = os.environ.get('MAGGIE_TOKEN_XBLOCK', '1') != '0'      # instance matte decoder: token-side levels of consecutive blocks in one launch


def token_linear_multi(layers):
    """`layers`: list of dicts with the keyword arguments of token_linear (x, W, b, xadd, res, relu, ln, wt) for layers that do NOT depend on each
    other -> list of outputs. One launch forward, two backward, for up to six layers (longer lists are cut into groups of six)."""
    outs = []
    for g0 in range(0, len(layers), TOKEN_MULTI_MAX):
        grp = layers[g0:g0 + TOKEN_MULTI_MAX]
        if not TOKEN_MULTI or len(grp) == 1:
            for L in grp:
                outs.append(token_linear(L['x'], L['W'], L.get('b'), L.get('xadd'), L.get('res'), L.get('relu', False), L.get('ln'), L.get('wt', False)))
            continue
        specs, slots = [], []
        for L in grp:
            ln = L.get('ln')
            specs.append((bool(L.get('relu', False)), ln.eps if ln is not None else 0.0, bool(L.get('wt', False))))
            slots += [L['x'], L.get('xadd'), L['W'], L.get('b'), L.get('res'), None if ln is None else ln.weight, None if ln is None else ln.bias]
        outs += list(TokenLinearMulti.apply(specs, *slots))
    return outs


class TokenSelfAttention(torch.autograd.Function):
    """softmax(q k^T / sqrt(d), key padding) v for (B, T <= 16, D) fp32 tokens: one workgroup per batch element each way."""

    @staticmethod
    def forward(ctx, q, k, v, pad):
        q, k, v = q.float().contiguous(), k.float().contiguous(), v.float().contiguous()
        B, T, D = q.shape
        pd = None if pad is None else as_u8(pad)
        out = torch.empty_like(q)
        prob = torch.empty((B, T, T), dtype=torch.float32, device=q.device)
        scale = 1.0 / (D ** 0.5)
        K.hip.call('mg_token_sa_fwd', K.hip.ptr(q), K.hip.ptr(k), K.hip.ptr(v), K.hip.ptr(pd), K.c_float(scale), K.c_int(B), K.c_int(T), K.c_int(D),
                   K.hip.ptr(out), K.hip.ptr(prob), K.hip.stream())
        ctx.save_for_backward(q, k, v, prob)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, prob = ctx.saved_tensors
        B, T, D = q.shape
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        K.hip.call('mg_token_sa_bwd', K.hip.ptr(dout.float().contiguous()), K.hip.ptr(q), K.hip.ptr(k), K.hip.ptr(v), K.hip.ptr(prob), K.c_float(ctx.scale),
                   K.c_int(B), K.c_int(T), K.c_int(D), K.hip.ptr(dq), K.hip.ptr(dk), K.hip.ptr(dv), K.hip.stream())
        return dq, dk, dv, None


def token_self_attention(q, k, v, pad=None):
    return TokenSelfAttention.apply(q, k, v, pad)


class RowsAddLayerNorm(torch.autograd.Function):
    """LayerNorm(x + r) over the channels of (rows, C) matrices (the post-norm residual of the feature-side cross attention over all
    b * L feature rows, mask_attention.py:128-133): mg_rows_add_layernorm_fwd / _bwd."""

    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps):
        shape = x.shape
        C = shape[-1]
        x2, r2 = x.contiguous().view(-1, C), r.contiguous().view(-1, C)
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        y, rstat = K.rows_add_layernorm(x2, r2, g, b, eps)
        ctx.save_for_backward(x2, r2, g, rstat)
        ctx.shape = shape
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, r2, g, rstat = ctx.saved_tensors
        dz, dg, db = K.rows_add_layernorm_bwd(dy.contiguous().view(x2.shape), x2, r2, g, rstat)
        dz = dz.view(ctx.shape)
        return dz, dz, dg, db, None


def rows_add_layernorm(x, r, ln):
    return RowsAddLayerNorm.apply(x, r, ln.weight, ln.bias, ln.eps)


class TokenEinsum(torch.autograd.Function):
    """logits (B, L, 16) = einsum('bqc,blc->blq') padded to 16 outputs: mg_token_einsum_fwd / _bwd (one launch each way)."""

    @staticmethod
    def forward(ctx, feat, tok):
        feat = feat.contiguous()
        tok32 = tok.detach().float().contiguous()
        B, L, C = feat.shape
        Q = tok32.shape[1]
        out = torch.empty((B, L, 16), dtype=feat.dtype, device=feat.device)
        K.hip.need_cuda(feat, tok32)
        K.hip.call('mg_token_einsum_fwd', K.hip.ptr(feat), K.c_int(K.hip.dtype_code(feat)), K.hip.ptr(tok32), K.c_int(B), K.c_int(L), K.c_int(C), K.c_int(Q),
                   K.c_int(16), K.hip.ptr(out), K.hip.stream())
        ctx.save_for_backward(feat, tok32)
        ctx.tok_dtype = tok.dtype
        return out

    @staticmethod
    def backward(ctx, dlog):
        feat, tok32 = ctx.saved_tensors
        B, L, C = feat.shape
        Q = tok32.shape[1]
        dlog = dlog.to(feat.dtype).contiguous()
        dfeat = torch.empty_like(feat)
        dtok = ARENA.acc(tok32.numel(), tok32.device).view_as(tok32)
        K.hip.call('mg_token_einsum_bwd', K.hip.ptr(dlog), K.hip.ptr(feat), K.c_int(K.hip.dtype_code(feat)), K.hip.ptr(tok32), K.c_int(B), K.c_int(L),
                   K.c_int(C), K.c_int(Q), K.c_int(16), K.hip.ptr(dfeat), K.hip.ptr(dtok), K.hip.stream())
        return dfeat, dtok if ctx.tok_dtype == torch.float32 else dtok.to(ctx.tok_dtype)


def token_einsum(feat, tok):
    """feat (B, L, C) in the compute dtype, tok (B, Q <= 16, C), C = 32 or 64 -> (B, L, 16) logits (columns >= Q are zero)."""
    return TokenEinsum.apply(feat, tok)
