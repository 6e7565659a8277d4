"""Device-resident sparse refinement head (detail stage of maggie/network/decoder/resnet_inst_matt_spconv.py:196-270).

The reference sizes every spconv feature matrix from `torch.nonzero` (:206) -- one device->host read per forward, after which the
whole head (~430 small launches forward + backward at the headline config) is paced by the host. Here nothing on the host ever
learns a site count:

  * `DevicePyramid`: the index pyramid OS1 -> OS2 -> OS4 -> OS8 (bit planes, ranks, sorted coordinates, gather tables) with the
    four site counts left in device words; coordinate lists / tables / feature matrices are allocated at their CAPACITY (every
    pixel of every plane active -- 288 GB of HBM is what makes that a non-issue) and only the first `count` rows are ever touched;
  * every kernel of the head takes the device word and runs over min(count, capacity) rows from a fixed grid (persistent
    implicit-GEMM tiles, grid-stride row kernels): `rows=` in maggie_amd.kernels;
  * `SparseHead` is ONE autograd.Function for the whole head: forward and backward are fixed launch sequences over preallocated
    row buffers (concatenations are channel slices of one buffer, a feature matrix with two consumers gets its gradient summed by
    a row kernel), so the detail stage has static shapes and is captured into hipGraphs next to the trunk (maggie_amd/graphs.py).

Numerically identical to the per-operator path of round 1 (same kernels, same order); BatchNorm over the active rows always uses the
exact two-pass variance (the row count is not known to the host, so there is no "few rows" special case to select)."""
import torch
import torch.distributed as dist

from . import functional as MF
from . import kernels as K
from .kernels import MODE_CONV, MODE_GATHER, ACT_NONE, ACT_RELU, ACT_LRELU

SLOPE = MF.LRELU_SLOPE


class DeviceLevel:
    """Active sites of one resolution level. `count` is a 1-element int32 DEVICE tensor (a view of the rank table's last entry)."""

    def __init__(self, bits, H, W, cap_frac=1.0, overflow=None, cap_abs=None):
        self.bits, self.H, self.W = bits, H, W
        self.P = bits.shape[0]
        full = self.P * H * W
        # capacity of every row buffer of this level. 1.0 (default): every site may be active, nothing can overflow. A smaller fraction bounds
        # the head's memory (dense-sized otherwise: ~50 GB at 512^2 x 40 planes); sites beyond it are dropped on the device and the sticky
        # `overflow` flag makes the model raise at its next flag read (MaGGIe._forward_impl)
        self.cap = full if (cap_frac >= 1.0 or overflow is None) else max(1024, int(full * cap_frac + 0.5))
        if cap_abs is not None and overflow is not None:      # an absolute row capacity (decoder.sparse_capacity 'auto': follows the workload's high-water mark)
            self.cap = max(1024, min(full, int(cap_abs)))
        self.rowoff, self.wordoff = K.bits_rank(bits, W)
        self.count = self.rowoff[-1:]
        if self.cap < full:
            K.bits_truncate_(bits, self.wordoff, W, self.cap, self.count, overflow)
        self.coords = None
        self._subm = None

    def finalize(self):
        self.coords = K.bits_coords(self.bits, self.wordoff, self.W, self.cap)

    def subm_table(self):
        if self._subm is None:
            self._subm = K.gather_table(self.coords, 3, 0, self.bits, self.wordoff, self.H, self.W, rows=self.count)
        return self._subm


class DevicePyramid:
    """OS1 / OS2 / OS4 / OS8 levels of the detail region (spconv SparseConv2d(k3,s2,p1) output rule), no host read.
    `patch`: (y0, y1, x0, x1) forced when the region is empty (training, resnet_inst_matt_spconv.py:347-348) -- decided on the device."""

    def __init__(self, roi_bits, H, W, patch=None, cap_frac=1.0, overflow=None, caps=None):
        if patch is not None:
            probe = K.bits_rank(roi_bits, W)[0][-1:]
            K.bits_patch_if_empty_(roi_bits, probe, H, W, *patch)
        # `cap_frac` is the OS1 fraction; a thin band of active OS1 sites covers a LARGER fraction of the coarser lattices (up to 4x per level,
        # about 2x for a band), so OS2 / OS4 / OS8 are sized for 2x / 4x / 8x the fraction (at most all sites): OS1 dominates the memory anyway
        # `caps`: absolute row capacities of the four levels instead of the fraction
        lv = [DeviceLevel(roi_bits, H, W, cap_frac, overflow, None if caps is None else caps[0])]
        for k in range(1, 4):
            b, h, w = K.bits_downsample(lv[-1].bits, lv[-1].W)
            lv.append(DeviceLevel(b, h, w, min(1.0, cap_frac * (2 ** k)), overflow, None if caps is None else caps[k]))
        for l in lv:
            l.finalize()
        self.levels = lv
        self._inv, self._down = {}, {}

    def inverse_tables(self, fine):
        """(fine<-coarse gather table over the fine rows, coarse<-fine table over the coarse rows) of SparseInverseConv2d at `fine`."""
        if fine not in self._inv:
            f, c = self.levels[fine], self.levels[fine + 1]
            self._inv[fine] = K.gather_table(f.coords, 3, 1, c.bits, c.wordoff, c.H, c.W, rows=f.count)
            self._down[fine] = K.gather_table(c.coords, 3, 2, f.bits, f.wordoff, f.H, f.W, rows=c.count)
        return self._inv[fine], self._down[fine]


class DeviceRng:
    """Counter-based dropout state on the device: int64 [seed, step]. `snapshot()` returns the state this forward uses and advances
    the step with a device op, so the sequence is identical whether the forward is launched eagerly or replayed from a hipGraph."""

    def __init__(self, device):
        self.state = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
        self._one = torch.tensor([0, 1], dtype=torch.int64, device=device)

    def snapshot(self):
        snap = self.state.clone()
        self.state.add_(self._one)
        return snap


def _gconv(x, w, nbr, rows, cap, bias=None, act=ACT_NONE, out=None, yoff=0, cout=None, xf=None):
    return K.conv_fprop(x, w, mode=MODE_GATHER, nbr=nbr, R=3, S=3, M=cap, shift=bias, act=act, pre_act=False, rows=rows, out=out, yoff=yoff,
                        cout=cout, xf=xf)


def _lin(x, w, rows, bias=None, relu=False, out=None, yoff=0, cout=None, xf=None):
    cap = x.shape[0]
    return K.conv_fprop(x, w, mode=MODE_CONV, N=1, Hin=1, Win=cap, Hout=1, Wout=cap, R=1, S=1, stride=1, pad=0, dil=1, shift=bias,
                        act=ACT_RELU if relu else ACT_NONE, pre_act=False, rows=rows, out=out, yoff=yoff, cout=cout, xf=xf)


def _wgrad_g(x, dy, cout, nbr, rows, dtype, park=None, xf=None):
    return K.conv_wgrad(x, dy, cout=cout, mode=MODE_GATHER, nbr=nbr, R=3, S=3, M=nbr.shape[0], out_dtype=dtype, rows=rows, park=park, xf=xf)


def _wgrad_l(x, dy, cout, rows, dtype, park=None, xf=None):
    cap = x.shape[0]
    return K.conv_wgrad(x, dy, cout=cout, mode=MODE_CONV, N=1, Hin=1, Win=cap, Hout=1, Wout=cap, R=1, S=1, stride=1, pad=0, dil=1,
                        out_dtype=dtype, rows=rows, park=park, xf=xf)


# The operand path for the head's BatchNorm1d layers is built and bit-exact (tests/test_gpu_conv.py::test_operand_transform_on_the_sparse_row_matrices,
# the model-level equality test runs with it on) but OFF by default: a gather 3x3 reads every neighbour row once PER TAP, so the transform runs nine times per
# element (the dense halo form transforms a staged pixel once) -- measured on the headline step: persistent gather kernels 0.635 -> 0.788 ms, per-tap weight
# gradients +0.03 ms, against 0.086 ms of apply passes saved (profiles/r05_trace_summary.txt vs r05_trace_summary_sparse_xf.txt). MAGGIE_LAZY_BN_SPARSE=1.
LAZY_SPARSE = __import__('os').environ.get('MAGGIE_LAZY_BN_SPARSE', '0') != '0'


class _BN:
    """BatchNorm1d over the live rows of a (capacity x C) matrix: training (batch statistics, running-stat update, optional SyncBN
    exchange) or eval (running statistics); keeps what its backward needs."""

    def __init__(self, bn, act):
        self.bn, self.act = bn, act
        self.training = bn.training or bn.running_mean is None

    def fwd(self, x, rows, gamma, beta, lazy=False):
        """-> the normalised, activated rows -- or, `lazy` (round 5: every consumer of this layer is a convolution whose kernels transform their
        operand in flight, csrc/conv_igemm.hip XF forms): x itself, with `self.xf` = (scale, shift, act, slope) for those consumers; the apply pass
        and the stored activation disappear, the backward re-forms the activation mask from x."""
        bn, C = self.bn, x.shape[1]
        self.x = x
        self.xf = None
        self.group = MF._sync_group(bn) if self.training else None
        if lazy and MF.LAZY_BN and LAZY_SPARSE and self.training and self.group is None and x.dtype != torch.float32:
            if bn.num_batches_tracked is not None and not MF.DEFER_BN_COUNTERS:
                if MF.BN_COUNT_LOG is not None:
                    MF.BN_COUNT_LOG.append(bn.num_batches_tracked)
                else:
                    bn.num_batches_tracked.add_(1)
            mom = 0.1 if bn.momentum is None else bn.momentum
            _, self.pack = K.bn_train_fwd(x, gamma, beta, bn.running_mean, bn.running_var, mom, bn.eps, self.act, SLOPE,
                                          stats_ws=MF.ARENA.take(K.stats_ws_floats(C, True), x.device), rows=rows, apply=False)
            self.y = None
            self.xf = (self.pack[:C], self.pack[C:2 * C], self.act, SLOPE)
            return x
        if self.training:
            if bn.num_batches_tracked is not None and not MF.DEFER_BN_COUNTERS:
                if MF.BN_COUNT_LOG is not None:
                    MF.BN_COUNT_LOG.append(bn.num_batches_tracked)
                else:
                    bn.num_batches_tracked.add_(1)
            mom = 0.1 if bn.momentum is None else bn.momentum
            if self.group is None:
                self.y, self.pack = K.bn_train_fwd(x, gamma, beta, bn.running_mean, bn.running_var, mom, bn.eps, self.act, SLOPE,
                                                   stats_ws=MF.ARENA.take(K.stats_ws_floats(C, True), x.device), rows=rows)
                return self.y
            # SyncBN: pooled moments of all ranks' live rows; the (variable) row count rides along in the pack
            from . import parallel as _par
            stats = MF.ARENA.zeros(K.stat_rows() * 2 * C, x.device).view(K.stat_rows(), 2 * C)
            K.hip.call('mg_colstats_dev', K.hip.ptr(x), K.c_int(K.hip.dtype_code(x)), K.c_int(x.shape[0]), K.c_int(C), K.c_int(C), K.hip.ptr(stats),
                       K.hip.ptr(rows), K.hip.stream())
            comm = _par.SYNCBN_COMM
            if comm is not None and hasattr(comm, 'bn_finalize') and comm.can_finalize(C) and x.is_cuda and rows.dtype == torch.int32:
                # mailbox exchange: ordered row sum, then exchange + finalize in ONE launch with the live-row count read from the device (was: cast +
                # cat + copy + exchange + finalize + cat)
                flat = K.stat_rows_sum(stats) if stats.shape[0] > K.STAT_REPLICAS else stats
                scale, shift, mean, invstd, self.cnt = comm.bn_finalize(flat, 0.0, gamma, beta, bn.running_mean, bn.running_var, mom, bn.eps, count_dev=rows)
                self.pack = scale._base
            else:
                pack = _par.syncbn_exchange_forward(torch.cat([K.stat_rows_sum(stats), rows.float()]), self.group)
                self.cnt = pack[2 * C:]
                scale, shift, mean, invstd = K.bn_finalize(pack[:2 * C], 0.0, gamma, beta, bn.running_mean, bn.running_var, mom, bn.eps, count_ptr=self.cnt)
                self.pack = torch.cat([scale, shift, mean, invstd])
        else:
            scale, shift = K.bn_fold(gamma, beta, bn.running_mean, bn.running_var, bn.eps)
            self.pack = torch.cat([scale, shift, bn.running_mean.float(), torch.rsqrt(bn.running_var.float() + bn.eps)])
        self.y = K.affine_act(x, self.pack[:C], self.pack[C:2 * C], act=self.act, slope=SLOPE, rows=rows)
        return self.y

    def bwd(self, dy, rows):
        """-> dx, dgamma, dbeta."""
        C = self.x.shape[1]
        capturing = torch.cuda.is_current_stream_capturing()
        if self.training and self.group is None:
            sums = MF.ARENA.take(2 * C, dy.device) if capturing else None
            dx, _, sums = K.bn_train_bwd(dy, self.y, self.x, self.pack, self.act, SLOPE, sums=sums, rows=rows)
            return dx, sums[C:], sums[:C]
        scale, mean, invstd = self.pack[:C], self.pack[2 * C:3 * C], self.pack[3 * C:]
        _, _, local = K.bn_backward(dy, self.y, self.x, scale, mean, invstd, 1.0, act=self.act, slope=SLOPE, reduce_only=True, rows=rows)
        if not self.training:                                     # eval statistics are constants: dx = g * scale
            zeros = torch.zeros(2 * C, dtype=torch.float32, device=dy.device)
            dx, _, _ = K.bn_backward(dy, self.y, self.x, scale, mean, invstd, 1.0, act=self.act, slope=SLOPE, sums=zeros, apply_only=True, rows=rows)
            return dx, local[C:], local[:C]
        from .parallel import syncbn_exchange_backward
        glob, local = syncbn_exchange_backward(local, self.group)
        dx, _, _ = K.bn_backward(dy, self.y, self.x, scale, mean, invstd, 1.0, act=self.act, slope=SLOPE, sums=glob, apply_only=True,
                                 count_ptr=self.cnt, rows=rows)
        return dx, local[C:], local[:C]


class SparseHead(torch.autograd.Function):
    """x_os4, x_os1 = head(os8_feat, tokens, fea1, fea2, fea3; weights) over the device pyramid `env.pyr` (see module docstring).

    `env`: namespace with .dec (the decoder module: BatchNorm / LayerNorm holders, dropout p), .pyr, .n_i, .rng_state (device int64[2]
    or None), .wb (list of (weight, bias) kernel-layout tensors in the order of `HEAD_CONVS`), and index lists into `params`."""

    @staticmethod
    def forward(ctx, env, os8_feat, tokens, fea1, fea2, fea3, *params):
        dec, pyr, n_i = env.dec, env.pyr, env.n_i
        l1, l2, l4, l8 = pyr.levels
        m1, m2, m4, m8 = l1.count, l2.count, l4.count, l8.count
        dt = os8_feat.dtype
        dev = os8_feat.device
        W = {name: params[i] for name, i in env.w_index.items()}
        Bv = {name: params[i] for name, i in env.b_index.items()}
        bnp = {name: (params[i], params[i + 1]) for name, i in env.bn_index.items()}
        ln_g, ln_b = params[env.ln_index], params[env.ln_index + 1]
        s = ctx.s = type('Saved', (), {})()                       # tensors kept for the backward live on this object
        s.env, s.dt, s.shapes = env, dt, (os8_feat.shape, fea1.shape, fea2.shape, fea3.shape, tokens.shape)
        inv4, dn8 = pyr.inverse_tables(2)
        inv2, dn4 = pyr.inverse_tables(1)
        inv1, dn2 = pyr.inverse_tables(0)
        t4, t1 = l4.subm_table(), l1.subm_table()
        s.tabs = (inv4, dn8, inv2, dn4, inv1, dn2, t4, t1)
        new = lambda cap, c: torch.empty((cap, c), dtype=dt, device=dev)   # noqa: E731
        tok32 = tokens.float().contiguous()
        s.tok32, s.os8_feat, s.tok_dtype, s.ln_g = tok32, os8_feat, tokens.dtype, ln_g
        # ---- OS8 rows * instance tokens -> inst_spec_layer (FFN, post-norm)  (:221-232) ----------------------------------------
        A0 = K.gather_rows(os8_feat, l8.coords, n_i, mul=tok32, rows=m8)
        ffn = dec.inst_spec_layer
        p_drop = ffn.dropout.p if ffn.training else 0.0
        s.p_drop, s.rng = p_drop, (env.rng_state if p_drop > 0 else None)
        H1 = _lin(A0, W['ffn1'], m8, Bv['ffn1'], relu=True)
        H1d = K.rows_dropout(H1, p_drop, s.rng, 1, rows=m8) if p_drop > 0 else H1
        H2 = _lin(H1d, W['ffn2'], m8, Bv['ffn2'])
        H2d = K.rows_dropout(H2, p_drop, s.rng, 2, rows=m8) if p_drop > 0 else H2
        A1, rstat = K.rows_add_layernorm(A0, H2d, ln_g, ln_b, ffn.norm.eps, rows=m8)
        s.A0, s.H1, s.H1d, s.H2d, s.rstat, s.A1 = A0, H1, H1d, H2d, rstat, A1
        # ---- layer3: inverse conv OS8 -> OS4, BN, LeakyReLU, SubM 3x3  (:69-74) ----------------------------------------------------
        bn = s.bn = {}

        def BN(name, x, rows, act, lazy=False):
            b = bn[name] = _BN(getattr(dec, name.split('.')[0])[int(name.split('.')[1])], act)
            return b.fwd(x, rows, *bnp[name], lazy=lazy)

        XF = lambda name: bn[name].xf                                     # noqa: E731  (None: the layer stored its output)

        B0 = _gconv(A1, W['layer3.0'], inv4, m4, l4.cap)
        B1 = BN('layer3.1', B0, m4, ACT_LRELU)
        CAT3 = new(l4.cap, 128)                                   # [fea3 rows | layer3 output]  (instance_spec_guidance, :172-194)
        K.gather_rows(fea3, l4.coords, n_i, out=CAT3, yoff=0, rows=m4)
        _gconv(B1, W['layer3.3'], t4, m4, l4.cap, out=CAT3, yoff=64)
        G0 = _lin(CAT3, W['guidance_layer.0'], m4)
        G1 = BN('guidance_layer.1', G0, m4, ACT_LRELU)
        G2 = _gconv(G1, W['guidance_layer.3'], t4, m4, l4.cap, bias=Bv['guidance_layer.3'])
        X4 = K.rows_sigmoid_mul(CAT3[:, :64], G2, rows=m4)
        S0 = _lin(X4, W['layer3_smooth.0'], m4, Bv['layer3_smooth.0'], relu=True)
        # (lazy: the BatchNorm1d output is never stored -- both consumers of S1, and the consumer of every layer marked below, are register-staged
        # gather / 1x1 kernels with Cout <= 32, which apply scale | shift | activation to their operand in flight)
        S1 = BN('layer3_smooth.2', S0, m4, ACT_NONE, lazy=True)
        R0 = _gconv(S1, W['refine_OS4.0'], t4, m4, l4.cap, xf=XF('layer3_smooth.2'))
        R1 = BN('refine_OS4.1', R0, m4, ACT_LRELU, lazy=True)
        R2 = _gconv(R1, W['refine_OS4.3'], t4, m4, l4.cap, bias=Bv['refine_OS4.3'], xf=XF('refine_OS4.1'))
        x_os4 = K.scatter_plane(R2, 0, l4.coords, l4.P, l4.H, l4.W, -99.0, rows=m4)
        s.CAT3, s.G2, s.X4, s.S0 = CAT3, G2, X4, S0
        # ---- layer4 (OS4 -> OS2), fea2, layer4_smooth  (:91-102) ----------------------------------------------------------------------
        C0 = _gconv(S1, W['layer4.0'], inv2, m2, l2.cap, xf=XF('layer3_smooth.2'))
        C1 = BN('layer4.1', C0, m2, ACT_LRELU, lazy=True)
        CAT2 = new(l2.cap, 64)
        K.gather_rows(fea2, l2.coords, n_i, out=CAT2, yoff=0, rows=m2)
        _lin(C1, W['layer4.3'], m2, out=CAT2, yoff=32, xf=XF('layer4.1'))
        D0 = _lin(CAT2, W['layer4_smooth.0'], m2, Bv['layer4_smooth.0'], relu=True)
        D1 = BN('layer4_smooth.2', D0, m2, ACT_NONE, lazy=True)
        s.CAT2, s.D0 = CAT2, D0
        # ---- layer5 (OS2 -> OS1), fea1, layer5_smooth, refine_OS1  (:105-130) -------------------------------------------------------
        E0 = _gconv(D1, W['layer5.0'], inv1, m1, l1.cap, xf=XF('layer4_smooth.2'))
        E1 = BN('layer5.1', E0, m1, ACT_LRELU, lazy=True)
        CAT1 = new(l1.cap, 64)
        K.gather_rows(fea1, l1.coords, n_i, out=CAT1, yoff=0, rows=m1)
        _gconv(E1, W['layer5.3'], t1, m1, l1.cap, out=CAT1, yoff=32, xf=XF('layer5.1'))
        F0 = _lin(CAT1, W['layer5_smooth.0'], m1, Bv['layer5_smooth.0'], relu=True)
        F1 = BN('layer5_smooth.2', F0, m1, ACT_NONE, lazy=True)
        Q0 = _gconv(F1, W['refine_OS1.0'], t1, m1, l1.cap, xf=XF('layer5_smooth.2'))
        Q1 = BN('refine_OS1.1', Q0, m1, ACT_LRELU, lazy=True)
        Q2 = _gconv(Q1, W['refine_OS1.3'], t1, m1, l1.cap, bias=Bv['refine_OS1.3'], xf=XF('refine_OS1.1'))
        x_os1 = K.scatter_plane(Q2, 0, l1.coords, l1.P, l1.H, l1.W, -99.0, rows=m1)
        s.CAT1, s.F0 = CAT1, F0
        s.W = W
        s.wt = {k: getattr(v, '_mg_wt', None) for k, v in W.items()}
        s.fea = (fea1, fea2, fea3)
        return x_os4, x_os1

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_os4, d_os1):
        s = ctx.s
        if s is None:
            raise K.hip.MaggieHipError('SparseHead: a second backward through the same forward (retain_graph=True) is not supported -- the head '
                                       'releases its saved row buffers after the first one')
        env, dt = s.env, s.dt
        pyr, n_i = env.pyr, env.n_i
        l1, l2, l4, l8 = pyr.levels
        m1, m2, m4, m8 = l1.count, l2.count, l4.count, l8.count
        inv4, dn8, inv2, dn4, inv1, dn2, t4, t1 = s.tabs
        W, bn = s.W, s.bn
        gW, gB, gBN = {}, {}, {}
        park = [] if MF.PARK_WGRAD else None                       # all the slab reductions of this backward run as ONE launch at its end

        def wt(name, reverse):
            t = s.wt[name]
            if t is None:
                t = W[name].permute(2, 1, 0)
                t = (t.flip(1) if reverse else t).contiguous()
            return t

        def bn_b(name, dy, rows):
            dx, dg, db = bn[name].bwd(dy.contiguous() if not dy.is_contiguous() else dy, rows)
            gBN[name] = (dg, db)
            return dx

        def src(b):
            """(x operand, transform) of a conv fed by BatchNorm layer `b`: its stored output, or -- operand-path layers -- its RAW input + xf."""
            return (b.y, None) if b.xf is None else (b.x, b.xf)

        def subm_b(name, x, dy, tab, rows, cap, cout, need_dx=True, xf=None):
            """SubM 3x3: dgrad = gather conv with the tap-reversed twin over the same table; wgrad over the table."""
            gW[name] = _wgrad_g(x, dy, cout, tab, rows, dt, park, xf=xf)
            return _gconv(dy, wt(name, True), tab, rows, cap) if need_dx else None

        def inv_b(name, x, dy, tab_f, tab_c, rows_f, rows_c, cap_c, cout, xf=None):
            """inverse conv (fine rows <- coarse rows): dgrad over the coarse rows with the strided table; wgrad over the fine rows."""
            gW[name] = _wgrad_g(x, dy, cout, tab_f, rows_f, dt, park, xf=xf)
            return _gconv(dy, wt(name, False), tab_c, rows_c, cap_c)

        def lin_b(name, x, dy, rows, cout, xf=None):
            gW[name] = _wgrad_l(x, dy, cout, rows, dt, park, xf=xf)
            return _lin(dy, wt(name, False), rows)

        fea1, fea2, fea3 = s.fea
        # ---- refine_OS1 ----------------------------------------------------------------------------------------------------------
        co = W['refine_OS1.3'].shape[0]
        dQ2 = K.gather_plane(d_os1.contiguous(), l1.coords, dt, width=co, rows=m1)
        _, gB['refine_OS1.3'] = K.bias_act_bwd(dQ2, None, True, rows=m1)
        b = bn['refine_OS1.1']
        dQ1 = subm_b('refine_OS1.3', src(b)[0], dQ2, t1, m1, l1.cap, co, xf=src(b)[1])
        dQ0 = bn_b('refine_OS1.1', dQ1, m1)
        bF = bn['layer5_smooth.2']
        dF1 = subm_b('refine_OS1.0', src(bF)[0], dQ0, t1, m1, l1.cap, 32, xf=src(bF)[1])
        dF0 = bn_b('layer5_smooth.2', dF1, m1)
        g, gB['layer5_smooth.0'] = K.bias_act_bwd(dF0, s.F0, True, rows=m1)
        dCAT1 = lin_b('layer5_smooth.0', s.CAT1, g, m1, 32)
        dfea1 = K.gather_rows_bwd_dense(dCAT1, l1.bits, l1.wordoff, n_i, fea1.shape, yoff=0)
        bE = bn['layer5.1']
        dE1 = subm_b('layer5.3', src(bE)[0], dCAT1[:, 32:], t1, m1, l1.cap, 32, xf=src(bE)[1])
        dE0 = bn_b('layer5.1', dE1, m1)
        bD = bn['layer4_smooth.2']
        dD1 = inv_b('layer5.0', src(bD)[0], dE0, inv1, dn2, m1, m2, l2.cap, 32, xf=src(bD)[1])
        # ---- layer4 ----------------------------------------------------------------------------------------------------------------
        dD0 = bn_b('layer4_smooth.2', dD1, m2)
        g, gB['layer4_smooth.0'] = K.bias_act_bwd(dD0, s.D0, True, rows=m2)
        dCAT2 = lin_b('layer4_smooth.0', s.CAT2, g, m2, 32)
        dfea2 = K.gather_rows_bwd_dense(dCAT2, l2.bits, l2.wordoff, n_i, fea2.shape, yoff=0)
        bC = bn['layer4.1']
        dC1 = lin_b('layer4.3', src(bC)[0], dCAT2[:, 32:], m2, 32, xf=src(bC)[1])
        dC0 = bn_b('layer4.1', dC1, m2)
        bS = bn['layer3_smooth.2']
        dS1_a = inv_b('layer4.0', src(bS)[0], dC0, inv2, dn4, m2, m4, l4.cap, 32, xf=src(bS)[1])
        # ---- refine_OS4 ------------------------------------------------------------------------------------------------------------
        co4 = W['refine_OS4.3'].shape[0]
        dR2 = K.gather_plane(d_os4.contiguous(), l4.coords, dt, width=co4, rows=m4)
        _, gB['refine_OS4.3'] = K.bias_act_bwd(dR2, None, True, rows=m4)
        bR = bn['refine_OS4.1']
        dR1 = subm_b('refine_OS4.3', src(bR)[0], dR2, t4, m4, l4.cap, co4, xf=src(bR)[1])
        dR0 = bn_b('refine_OS4.1', dR1, m4)
        dS1_b = subm_b('refine_OS4.0', src(bS)[0], dR0, t4, m4, l4.cap, 32, xf=src(bS)[1])
        dS1 = K.rows_add(dS1_a, dS1_b, out=dS1_a, rows=m4)        # S1 feeds layer4 AND refine_OS4
        dS0 = bn_b('layer3_smooth.2', dS1, m4)
        g, gB['layer3_smooth.0'] = K.bias_act_bwd(dS0, s.S0, True, rows=m4)
        dX4 = lin_b('layer3_smooth.0', s.X4, g, m4, 64)
        # ---- instance-specific guidance ------------------------------------------------------------------------------------------
        d_detail, dG2 = K.rows_sigmoid_mul_bwd(dX4, s.CAT3[:, :64], s.G2, rows=m4)
        _, gB['guidance_layer.3'] = K.bias_act_bwd(dG2, None, True, rows=m4)
        bG = bn['guidance_layer.1']
        dG1 = subm_b('guidance_layer.3', bG.y, dG2, t4, m4, l4.cap, 64)
        dG0 = bn_b('guidance_layer.1', dG1, m4)
        dCAT3 = lin_b('guidance_layer.0', s.CAT3, dG0, m4, 64)     # (cap4, 128)
        K.rows_add(dCAT3[:, :64], d_detail, out=dCAT3[:, :64], rows=m4)      # fea3 rows feed the concat AND the product
        dfea3 = K.gather_rows_bwd_dense(dCAT3, l4.bits, l4.wordoff, n_i, fea3.shape, yoff=0)
        bB = bn['layer3.1']
        dB1 = subm_b('layer3.3', bB.y, dCAT3[:, 64:], t4, m4, l4.cap, 64)
        dB0 = bn_b('layer3.1', dB1, m4)
        dA1 = inv_b('layer3.0', s.A1, dB0, inv4, dn8, m4, m8, l8.cap, 64)
        # ---- inst_spec_layer ---------------------------------------------------------------------------------------------------------
        dz, d_ln_g, d_ln_b = K.rows_add_layernorm_bwd(dA1, s.A0, s.H2d, s.ln_g, s.rstat, rows=m8)
        dH2 = K.rows_dropout(dz, s.p_drop, s.rng, 2, rows=m8) if s.p_drop > 0 else dz
        _, gB['ffn2'] = K.bias_act_bwd(dH2, None, True, rows=m8)
        dH1d = lin_b('ffn2', s.H1d, dH2, m8, 64)
        dH1 = K.rows_dropout(dH1d, s.p_drop, s.rng, 1, rows=m8) if s.p_drop > 0 else dH1d
        g, gB['ffn1'] = K.bias_act_bwd(dH1, s.H1, True, rows=m8)
        dA0 = lin_b('ffn1', s.A0, g, m8, 64)
        dA0 = K.rows_add(dA0, dz, out=dA0, rows=m8)               # A0 feeds the FFN AND the residual
        d_os8 = K.gather_rows_bwd_dense(dA0, l8.bits, l8.wordoff, n_i, s.os8_feat.shape, mul=s.tok32)
        _, dtok = K.gather_rows_bwd(dA0, l8.coords, n_i, s.os8_feat.shape, mul=s.tok32, dense=s.os8_feat, want_ddense=False, want_dmul=True, rows=m8)
        if park:
            K.wgrad_reduce_batched(park)
        # ---- hand the gradients back in the order of `params` ---------------------------------------------------------------------------
        n_par = env.n_params
        out = [None] * n_par
        for name, i in env.w_index.items():
            out[i] = gW.get(name)
        for name, i in env.b_index.items():
            out[i] = gB.get(name)
        for name, i in env.bn_index.items():
            dg, db = gBN[name]
            out[i], out[i + 1] = dg, db
        out[env.ln_index], out[env.ln_index + 1] = d_ln_g, d_ln_b
        ctx.s = None
        return (None, d_os8, dtok.to(s.tok_dtype), dfea1, dfea2, dfea3) + tuple(out)
