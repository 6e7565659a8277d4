"""ctypes binding of the C ABI in include/maggie_hip.h (libmaggie_hip.so, hand-written HIP for gfx950).

PyTorch is only plumbing here: it owns device memory and streams; every call below hands raw device pointers,
sizes and the current HIP stream to the C ABI. There is NO fallback: if the shared library is missing or fails to
load, importing a kernel raises -- the product path never silently routes through PyTorch/CPU code.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MAGGIE_LIB_PATH') or os.path.join(_HERE, 'libmaggie_hip.so')      # (MAGGIE_LIB_PATH: an A/B build of the same ABI, tools/build_variant.sh)
_LIB = None
# Bit-reproducible steps (default): the reference runs with torch.backends.cudnn.deterministic = True (tools/main.py:135-136). Every cross-workgroup
# fp32 sum of the library then runs as "one partial per workgroup, added in index order" instead of atomicAdd (csrc/det.hip). MAGGIE_DETERMINISTIC=0
# switches back to the atomic forms.
DETERMINISTIC = os.environ.get('MAGGIE_DETERMINISTIC', '1') != '0'
DET_SCRATCH_BYTES = int(os.environ.get('MAGGIE_DET_SCRATCH_MB', '64')) << 20
_DET_READY = set()          # device indices whose slot scratch exists
_HAS_GPU = torch.cuda.is_available()

F32, BF16, F16 = 0, 1, 3          # MG_F32 / MG_BF16 / MG_F16 (2 is MG_U8, mask planes only)
ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
MODE_CONV, MODE_TCONV, MODE_GATHER = 0, 1, 2


class MaggieHipError(RuntimeError):
    pass


class ConvParams(ctypes.Structure):
    _fields_ = [
        ('x', ctypes.c_void_p), ('w', ctypes.c_void_p), ('y', ctypes.c_void_p), ('nbr', ctypes.c_void_p),
        ('scale', ctypes.c_void_p), ('shift', ctypes.c_void_p), ('res', ctypes.c_void_p), ('res2', ctypes.c_void_p),
        ('stats', ctypes.c_void_p),
        ('dtype', ctypes.c_int32), ('mode', ctypes.c_int32),
        ('N', ctypes.c_int32), ('Hin', ctypes.c_int32), ('Win', ctypes.c_int32), ('Cin', ctypes.c_int32),
        ('Hout', ctypes.c_int32), ('Wout', ctypes.c_int32), ('Cout', ctypes.c_int32),
        ('R', ctypes.c_int32), ('S', ctypes.c_int32), ('stride', ctypes.c_int32), ('pad', ctypes.c_int32),
        ('dil', ctypes.c_int32), ('M', ctypes.c_int32),
        ('ldx', ctypes.c_int32), ('ldy', ctypes.c_int32), ('yoff', ctypes.c_int32), ('ldr', ctypes.c_int32),
        ('ldr2', ctypes.c_int32),
        ('act', ctypes.c_int32), ('pre_act', ctypes.c_int32), ('res_mode', ctypes.c_int32),
        ('slope', ctypes.c_float), ('dw_dtype', ctypes.c_int32), ('stat_mode', ctypes.c_int32),
        ('m_dev', ctypes.c_void_p),
        ('bnb_y', ctypes.c_void_p), ('bnb_x', ctypes.c_void_p), ('bnb_mean', ctypes.c_void_p), ('bnb_invstd', ctypes.c_void_p),
        ('bnb_act', ctypes.c_int32), ('bnb_ld', ctypes.c_int32),
        ('stat_rep', ctypes.c_int32), ('reserved0', ctypes.c_int32),
        ('xf_scale', ctypes.c_void_p), ('xf_shift', ctypes.c_void_p), ('xf_act', ctypes.c_int32), ('xf_slope', ctypes.c_float),
        ('bnb_scale', ctypes.c_void_p), ('bnb_shift', ctypes.c_void_p),
    ]


def lib():
    """Load libmaggie_hip.so (built in-tree by __graft_entry__.build()); fail loudly if it is not there."""
    global _LIB
    if _LIB is None:
        if not os.path.isfile(LIB_PATH):
            raise MaggieHipError(
                'libmaggie_hip.so not found at %s -- run `python -c "import __graft_entry__ as g; g.build()"` '
                '(hipcc --offload-arch=gfx950). There is no CPU / PyTorch fallback for the MaGGIe hot path.' % LIB_PATH)
        _LIB = ctypes.CDLL(LIB_PATH)
        _LIB.mg_abi_version.restype = ctypes.c_int
        _LIB.mg_set_deterministic(ctypes.c_int(int(DETERMINISTIC)))
    return _LIB


def set_deterministic(on):
    """Switch the library between the ordered-slot sums (bit-reproducible, default) and the atomic forms. Graphs captured before the switch keep
    the kernels they recorded; buffers sized by the old mode (kernels.STAT_ROWS) must not be reused."""
    global DETERMINISTIC
    if on:
        # the same guard as at import time (functional.py): library kernels on concurrent streams would share the slot scratch
        from . import functional as _MF
        if _MF.SIDE_WGRAD or _MF.PAR_BRANCHES != '0':
            raise MaggieHipError('set_deterministic(True) with MAGGIE_SIDE_WGRAD / MAGGIE_BRANCHES: library kernels on concurrent streams share the '
                                 'slot scratch of the deterministic sums')
    DETERMINISTIC = bool(on)
    lib().mg_set_deterministic(ctypes.c_int(int(DETERMINISTIC)))


def _det_init():
    """The library's slot scratch lives on the device the process computes on: allocated at the first call into the library (never inside a
    stream capture -- every captured path runs eagerly first)."""
    if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
        check(lib().mg_det_init(ctypes.c_long(DET_SCRATCH_BYTES)), 'mg_det_init')
        _DET_READY.add(torch._C._cuda_getDevice())


def code_of(dtype):
    """torch dtype -> MG_* code (fp32, bf16, fp16)."""
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float16:
        return F16
    raise MaggieHipError('unsupported dtype %s' % dtype)


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise MaggieHipError('unsupported dtype %s' % t.dtype)


def ptr(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """The hipStream_t torch is currently launching on (raw query: torch.cuda.current_stream() builds a Python Stream object
    per call, ~5 us, and this runs once per kernel launch)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch._C._cuda_getDevice()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_RC_TEXT = {
    -7: 'the slot scratch of the deterministic sums is missing or too small for this launch -- it is allocated once per device and never moved '
        '(captured graphs hold its address): start the process with a larger MAGGIE_DET_SCRATCH_MB (now %d)' % (DET_SCRATCH_BYTES >> 20),
    -8: 'the BatchNorm statistics buffer has fewer rows than the producing kernel form has output tiles (deterministic mode: one row per tile, '
        'mg_conv_stat_rows)',
}


def check(rc, what):
    if rc != 0:
        raise MaggieHipError('%s failed with code %d%s' % (what, rc, (': ' + _RC_TEXT[rc]) if rc in _RC_TEXT else ''))


def need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise MaggieHipError('MaGGIe HIP kernels need device tensors (got a CPU tensor); there is no CPU fallback')


_TIMING = None          # set by enable_timing(): {'names': set, 'records': {name: [(start, end, work, tag)]}}


def enable_timing(names):
    """bench.py instrumentation: bracket every launch of the named entry points with HIP events on the launch stream
    (torch.cuda.Event records on torch's current stream, which is the stream the kernels are launched on)."""
    global _TIMING
    _TIMING = {'names': set(names), 'records': {n: [] for n in names}}


def disable_timing():
    global _TIMING
    t, _TIMING = _TIMING, None
    return t


_FN = {}
_SYNC_CALLS = os.environ.get('MAGGIE_SYNC_CALLS', '0') == '1'      # debugging: synchronise after every launch, so that an asynchronous GPU fault names its entry point


def call(name, *args, work=None, tag=None):
    if _SYNC_CALLS and not torch.cuda.is_current_stream_capturing():
        import sys
        sys.stderr.write('[mg] %s\n' % name)
        sys.stderr.flush()
        _call(name, *args, work=work, tag=tag)
        torch.cuda.synchronize()
        return
    _call(name, *args, work=work, tag=tag)


def _call(name, *args, work=None, tag=None):
    if _HAS_GPU and torch._C._cuda_getDevice() not in _DET_READY:
        _det_init()
    fn = _FN.get(name)
    if fn is None:
        fn = getattr(lib(), name)
        fn.restype = ctypes.c_int
        _FN[name] = fn
    if _TIMING is not None and name in _TIMING['names']:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*args)
        e.record()
        _TIMING['records'][name].append((s, e, work, tag))
        check(rc, name)
        return
    check(fn(*args), name)


class RowwiseParams(ctypes.Structure):
    _fields_ = [
        ('x', ctypes.c_void_p), ('y', ctypes.c_void_p), ('dy', ctypes.c_void_p), ('dx', ctypes.c_void_p),
        ('res', ctypes.c_void_p), ('res2', ctypes.c_void_p), ('dres', ctypes.c_void_p),
        ('scale', ctypes.c_void_p), ('shift', ctypes.c_void_p), ('mean', ctypes.c_void_p), ('invstd', ctypes.c_void_p),
        ('sums', ctypes.c_void_p), ('count_ptr', ctypes.c_void_p),
        ('dtype', ctypes.c_int32), ('M', ctypes.c_int32), ('C', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32),
        ('ldx', ctypes.c_int32), ('ldy', ctypes.c_int32), ('yoff', ctypes.c_int32), ('ldr', ctypes.c_int32),
        ('ldr2', ctypes.c_int32), ('lddy', ctypes.c_int32), ('lddx', ctypes.c_int32), ('lddres', ctypes.c_int32),
        ('act', ctypes.c_int32), ('res_mode', ctypes.c_int32), ('mask_x_pos', ctypes.c_int32),
        ('slope', ctypes.c_float), ('count', ctypes.c_float),
        ('m_dev', ctypes.c_void_p),
        ('count_mult', ctypes.c_int32), ('mask_from_x', ctypes.c_int32),
    ]


U8 = 2
c_int, c_float, c_long = ctypes.c_int, ctypes.c_float, ctypes.c_long


class TokLin(ctypes.Structure):
    """mg_tok_lin (include/maggie_hip.h): one layer of mg_token_linear_multi_fwd / _bwd."""
    _fields_ = [(n, ctypes.c_void_p) for n in ('x', 'xadd', 'W', 'bias', 'res', 'gamma', 'beta', 'y', 'z', 'rstat', 'dy', 'yout', 'dx', 'dW', 'db',
                                               'dres', 'dgamma', 'dbeta', 'dz')] + \
               [(n, ctypes.c_int32) for n in ('R', 'K', 'N', 'relu', 'wt')] + [('eps', ctypes.c_float), ('dx_pair', ctypes.c_int32)]


class WbEntry(ctypes.Structure):
    """mg_wb_entry (include/maggie_hip.h)."""
    _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('dst_t', ctypes.c_void_p), ('cout', ctypes.c_int32),
                ('taps', ctypes.c_int32), ('cin', ctypes.c_int32), ('cout_pad', ctypes.c_int32), ('cin_pad', ctypes.c_int32),
                ('flip_t', ctypes.c_int32), ('dtype', ctypes.c_int32), ('reserved', ctypes.c_int32)]


class WgradParked(ctypes.Structure):
    """mg_wgrad_parked (include/maggie_hip.h): a weight-gradient slab reduction that was not launched yet."""
    _fields_ = [('ws', ctypes.c_void_p), ('dw', ctypes.c_void_p), ('n', ctypes.c_long), ('splits', ctypes.c_int32), ('form', ctypes.c_int32),
                ('dw_dtype', ctypes.c_int32), ('blocks', ctypes.c_int32)]


class SnDesc(ctypes.Structure):
    _fields_ = [('W', ctypes.c_void_p), ('u', ctypes.c_void_p), ('v', ctypes.c_void_p), ('out_off', ctypes.c_int64),
                ('work_off', ctypes.c_int64), ('dw_off', ctypes.c_int64), ('A', ctypes.c_int32), ('B', ctypes.c_int32),
                ('taps', ctypes.c_int32), ('transposed', ctypes.c_int32), ('pad_in', ctypes.c_int32), ('plain', ctypes.c_int32),
                ('k3_first', ctypes.c_int32), ('k3_count', ctypes.c_int32)]
