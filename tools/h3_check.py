"""halo3 (csrc/conv_halo3.hip) against the round-2 halo kernels (same entry point, mg_set_halo3(0)) and against a torch fp32 convolution of the
same rounded operands: correctness over the epilogue features, then per-shape timing (back-to-back launches) old | new [| forced tile forms].
usage: python tools/h3_check.py [check] [time] [cfgs]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from maggie_amd import hip, kernels as K
dev = torch.device('cuda:0')
lib = hip.lib()


def h3(on):
    lib.mg_set_halo3(ctypes.c_int(int(on)))


def cfg(th=0, bn=0, ns=0):
    lib.mg_set_halo3_cfg(ctypes.c_int(th), ctypes.c_int(bn), ctypes.c_int(ns))


def bench(fn, it=20, reps=10):
    """per-launch period of `it` back-to-back launches replayed from a hipGraph (the Python binding costs ~10 us per call: eager
    back-to-back timing measures the host for any kernel shorter than that)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(it): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (it * reps) * 1e3


def ref_conv(x, w, N, H, W, mode):
    """x (N*H*W, Cin) bf16, w (Cout, 9, Cin): fp32 conv of the rounded operands; TCONV = taps mirrored (stride-1 data gradient form)"""
    Cin, Cout = x.shape[1], w.shape[0]
    xi = x.float().view(N, H, W, Cin).permute(0, 3, 1, 2)
    wk = w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    if mode == K.MODE_TCONV:
        wk = wk.flip(2, 3)
    y = F.conv2d(xi, wk, padding=1)
    return y.permute(0, 2, 3, 1).reshape(N * H * W, Cout)


def check():
    torch.manual_seed(0)
    if os.environ.get('H3_CHECK_CFG'):                        # one tile form for every halo3 launch of the check (e.g. 8,64,103: the persistent ring form)
        cfg(*[int(v) for v in os.environ['H3_CHECK_CFG'].split(',')])
    cases = []
    for (N, Cin, Cout, H, W) in [(2, 128, 128, 16, 32), (1, 64, 64, 24, 40), (2, 32, 32, 20, 36), (1, 256, 96, 8, 16), (3, 96, 48, 12, 20), (1, 512, 64, 5, 17),
                                 (4, 128, 128, 64, 64), (1, 160, 32, 9, 33),
                                 # one slab, one channel tile: the persistent weights-once form (conv_halo3_slab_kernel) -- several tiles per workgroup, ragged edges, Cout 8 .. 32
                                 (4, 32, 32, 128, 128), (1, 32, 8, 40, 40), (2, 32, 16, 24, 56), (1, 32, 24, 9, 33), (4, 32, 8, 128, 128), (3, 32, 32, 200, 72)]:
        for mode in (K.MODE_CONV, K.MODE_TCONV):
            cases.append((N, Cin, Cout, H, W, mode))
    bad = 0
    for (N, Cin, Cout, H, W, mode) in cases:
        M = N * H * W
        x = torch.randn(M, Cin, device=dev).bfloat16()
        w = (torch.randn(Cout, 9, Cin, device=dev) / (9 * Cin) ** 0.5).bfloat16()
        scale = torch.rand(Cout, device=dev) + 0.5
        shift = torch.randn(Cout, device=dev)
        res = torch.randn(M, Cout, device=dev).bfloat16()
        res_half = torch.randn(N * (H // 2) * (W // 2), Cout, device=dev).bfloat16() if H % 2 == 0 and W % 2 == 0 else None
        res2 = torch.randn(M, Cout, device=dev).bfloat16()
        geo = dict(N=N, Hin=H, Win=W, R=3, S=3, stride=1, pad=1, dil=1, mode=mode)
        rows = K.conv_stat_rows(M, N, H, W)
        variants = [dict(), dict(scale=scale, shift=shift, act=K.ACT_RELU), dict(shift=shift, act=K.ACT_LRELU, slope=0.2, res=res),
                    dict(scale=scale, shift=shift, act=K.ACT_RELU, pre_act=True, res2=res2), dict(scale=scale, shift=shift, act=K.ACT_RELU, res=res, res2=res2)]
        if res_half is not None:
            variants.append(dict(scale=scale, shift=shift, res=res_half, res_mode=2))
        for vi, kw in enumerate(variants):
            outs = []
            for on in (0, 1):
                h3(on)
                st = torch.zeros(rows, 2 * Cout, device=dev)
                wide = torch.zeros(M, Cout + 16, device=dev).bfloat16()           # the output lands in a channel slice of a wider buffer
                y = K.conv_fprop(x, w, stats=st, out=wide, yoff=8, cout=Cout, **geo, **kw)
                torch.cuda.synchronize()
                outs.append((wide.clone(), st.sum(0)))
            (yo, so), (yn, sn) = outs
            r = ref_conv(x, w, N, H, W, mode)
            sl = 1.0 if kw.get('act', K.ACT_NONE) == K.ACT_NONE else (0.0 if kw['act'] == K.ACT_RELU else kw.get('slope', 0.2))
            act = lambda v: torch.maximum(v, v * sl)
            if kw.get('pre_act'):
                r = act(r)
            r = r * kw.get('scale', torch.ones_like(scale)) + kw.get('shift', torch.zeros_like(shift))
            if 'res' in kw:
                rr = kw['res'].float()
                if kw.get('res_mode') == 2:
                    rr = rr.view(N, H // 2, W // 2, Cout).repeat_interleave(2, 1).repeat_interleave(2, 2).reshape(M, Cout)
                r = r + rr
            if not kw.get('pre_act'):
                r = act(r)
            if 'res2' in kw:
                r = r + kw['res2'].float()
            yn_f = yn[:, 8:8 + Cout].float()
            e_ref = (yn_f - r).abs().max().item() / max(1.0, r.abs().max().item())
            e_old = (yn.float() - yo.float()).abs().max().item()
            pad_ok = bool((yn[:, :8] == 0).all() and (yn[:, 8 + Cout:] == 0).all())
            rs = torch.cat([yn_f.sum(0), (yn_f * yn_f).sum(0)])
            e_st = ((sn - rs).abs() / (rs.abs() + 1.0)).max().item()
            ok = e_ref < 1e-2 and pad_ok and e_st < 2e-3
            bad += not ok
            print('%s N%d C%d->%d %dx%d mode %d variant %d: vs torch %.2e  vs old kernel %.2e  stats %.2e  slice intact %s' % (
                'ok  ' if ok else 'FAIL', N, Cin, Cout, H, W, mode, vi, e_ref, e_old, e_st, pad_ok))
    # operand transform (xf): the raw producer output + (scale, shift, act) must give the bits of the stored z = affine_act(y)
    for (N, Cin, Cout, H, W) in [(2, 32, 32, 20, 24), (4, 32, 32, 128, 128), (1, 32, 8, 33, 47), (3, 32, 32, 200, 72), (2, 32, 64, 20, 24), (1, 64, 32, 9, 17), (2, 128, 64, 16, 16), (1, 96, 64, 16, 24), (1, 256, 128, 8, 16), (1, 512, 64, 8, 16), (4, 128, 128, 64, 64), (4, 64, 64, 128, 128)]:
        for act in (0, 1, 2):
            M = N * H * W
            y = (torch.randn(M, Cin, device=dev) * 1.5 + 0.3).bfloat16()
            sc = (torch.rand(Cin, device=dev) + 0.5) * torch.where(torch.rand(Cin, device=dev) < 0.5, -1.0, 1.0)
            sh = torch.randn(Cin, device=dev) * 0.7
            w = (torch.randn(Cout, 9, Cin, device=dev) / (9 * Cin) ** 0.5).bfloat16()
            z = K.affine_act(y, sc, sh, act=act, slope=0.2)
            geo = dict(mode=K.MODE_CONV, N=N, Hin=H, Win=W, R=3, S=3, stride=1, pad=1, dil=1)
            rows = K.conv_stat_rows(M, N, H, W)
            h3(1)
            sa, sb = torch.zeros(rows, 2 * Cout, device=dev), torch.zeros(rows, 2 * Cout, device=dev)
            o_ref = K.conv_fprop(z, w, stats=sa, **geo)
            try:
                o_xf = K.conv_fprop(y, w, stats=sb, xf=(sc, sh, act, 0.2), **geo)
            except Exception as e:                            # (-9: no kernel form of the current dispatch transforms this geometry, e.g. Cout 8 outside the slab form)
                print('skip xf N%d C%d->%d %dx%d act %d: %s' % (N, Cin, Cout, H, W, act, str(e)[-40:]))
                continue
            h3(0)
            try:
                o_old = K.conv_fprop(y, w, xf=(sc, sh, act, 0.2), **geo)
            except Exception:                                 # (a geometry the round-5 transform kernels do not take)
                o_old = o_xf
            ok = torch.equal(o_ref, o_xf) and torch.equal(sa, sb)
            bad += not ok
            print('%s xf N%d C%d->%d %dx%d act %d: %d of %d values differ from the stored form; vs old xf kernel %.2e' % (
                'ok  ' if ok else 'FAIL', N, Cin, Cout, H, W, act, int((o_ref != o_xf).sum()), o_ref.numel(), (o_xf.float() - o_old.float()).abs().max().item()))
    print('FAILED: %d' % bad if bad else 'all cases ok')
    return bad


SHAPES_ALL = [(4, 32, 8, 512, 1), (12, 32, 32, 512, 0), (12, 32, 32, 256, 1), (12, 128, 128, 64, 1), (12, 256, 128, 64, 0), (32, 128, 128, 64, 0), (32, 256, 256, 32, 0), (4, 32, 32, 512, 0), (4, 32, 32, 512, 1), (4, 32, 32, 256, 0), (4, 32, 64, 128, 0), (4, 64, 64, 128, 0), (4, 64, 64, 128, 1), (4, 128, 128, 64, 0), (4, 128, 128, 64, 1),
          (4, 256, 256, 32, 0), (4, 256, 256, 32, 1), (4, 512, 512, 16, 0), (4, 512, 256, 32, 0), (4, 256, 128, 64, 0), (12, 128, 128, 64, 0), (12, 256, 256, 32, 0), (12, 64, 64, 128, 0)]


def timing(cfgs):
    tot = {}
    sel = os.environ.get('H3_SHAPES')
    shapes = [sh for sh in SHAPES_ALL if not sel or ('%d@%d' % (sh[1], sh[3])) in sel.split(',')]
    for (N, Cin, Cout, HW, mode) in shapes:
        NSET = 4                                                 # operand sets in rotation: a launch does not find its own inputs hot in L2
        xs = [torch.randn(N * HW * HW, Cin, device=dev).bfloat16() for _ in range(NSET)]
        ws = [(torch.randn(Cout, 9, Cin, device=dev) / (9 * Cin) ** 0.5).bfloat16() for _ in range(NSET)]
        ys = [torch.empty(N * HW * HW, Cout, device=dev).bfloat16() for _ in range(NSET)]
        st = torch.zeros(K.conv_stat_rows(N * HW * HW, N, HW, HW), 2 * Cout, device=dev)
        cnt = [0]

        def run():
            i = cnt[0] % NSET
            cnt[0] += 1
            K.conv_fprop(xs[i], ws[i], out=ys[i], stats=st if mode == 0 else None, **geo)
        geo = dict(N=N, Hin=HW, Win=HW, R=3, S=3, stride=1, pad=1, dil=1, mode=mode)
        fl = 2.0 * N * HW * HW * Cin * Cout * 9
        line = 'N%-2d C%d->%d %dx%d mode %d:' % (N, Cin, Cout, HW, HW, mode)
        for name, c in [('old', None), ('new', (0, 0, 0))] + [('%d,%d,%d' % c_, c_) for c_ in cfgs]:
            h3(c is not None)
            cfg(*(c or (0, 0, 0)))
            try:
                cnt[0] = 0
                tf = bench(run)
            except Exception as e:
                line += '  %s: n/a' % name
                continue
            tot[name] = tot.get(name, 0.0) + (tf if N == 4 else 0.0)
            line += '  %s %.1f us (%.0f TF)' % (name, tf, fl / tf / 1e6)
        print(line, flush=True)
    cfg()
    print('batch-4 totals: ' + '  '.join('%s %.1f us' % kv for kv in tot.items()))


if __name__ == '__main__':
    args = sys.argv[1:] or ['check', 'time']
    rc = 0
    if 'check' in args:
        rc = check()
    if 'time' in args:
        cfgs = [tuple(int(v) for v in a.split(',')) for a in args if ',' in a]
        timing(cfgs)
    sys.exit(1 if rc else 0)
