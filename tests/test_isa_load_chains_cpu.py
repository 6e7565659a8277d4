"""ISA-level regression checks that need no GPU (hipcc cross-compiles gfx950 here): the properties DESIGN.md 11.13 is about are visible in the
compiler's output, so they are pinned there -- the per-thread accumulators of the mask-embedding backward stay out of scratch memory, and the
batched-load stencils keep their loads in front of the arithmetic (few FULL waits for many loads). tools/isa_load_chains.py does the counting."""
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def _isa(src):
    import isa_load_chains as ilc
    with tempfile.NamedTemporaryFile(suffix='.s') as f:
        subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', os.path.join(ROOT, 'maggie_amd', 'csrc', src), '-o', f.name],
                       check=True, stderr=subprocess.DEVNULL, timeout=600)
        asm = open(f.name).read()
    out = {}
    for name, loads, full, waits, scratch in ilc.kernels(asm):
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        out[dem] = (loads, full, waits, scratch)
    return out


@pytest.mark.skipif(not os.path.isfile(HIPCC) or shutil.which('c++filt') is None, reason='hipcc / c++filt not available')
def test_batched_load_kernels_keep_their_shape_in_the_isa():
    with ThreadPoolExecutor(2) as ex:
        sparse, losses = ex.map(_isa, ['sparse.hip', 'losses.hip'])

    def pick(table, key):
        hits = {k: v for k, v in table.items() if key in k}
        assert hits, key
        return hits

    # registers, not scratch: an early `break` in the plane loop once kept it from unrolling and put the [16][3] accumulators into scratch memory
    for name, (loads, full, waits, scratch) in pick(sparse, 'mask_embed_bwd_det_kernel').items():
        assert scratch == 0, (name, scratch)
        assert full <= loads // 2, (name, loads, full)
    # no kernel of the two files may spill or index registers dynamically
    for table in (sparse, losses):
        for name, (_, _, _, scratch) in table.items():
            assert scratch == 0, (name, scratch)
    # the quad / window stencils: tens of loads, a handful of full waits (the tap-by-tap forms have one full wait per load)
    for key in ('pyr_lap_fwd_quad_kernel', 'pyr_upT_batched_kernel', 'pyr_downT_quad_kernel'):         # (point_bwd keeps a tap-by-tap border walk)
        for name, (loads, full, waits, scratch) in pick(losses, key).items():
            assert loads >= 20 and full <= loads // 2, (name, loads, full)
    for key in ('pyr_lap_fwd_kernel', 'pyr_downT_kernel'):          # the first forms, kept behind MG_LOSS_BATCHED=0: the contrast the rewrite was about
        for name, (loads, full, waits, scratch) in pick(losses, '::' + key + '(').items():
            assert full >= loads * 3 // 4, (name, loads, full)


def test_inflight_lds_checker_sees_a_count_that_is_too_generous():
    """tools/isa_lds_inflight.py on a hand-written stretch: three reads, a wait that leaves two outstanding, then uses of the first (landed) and of the
    second (may be in flight); a label clears the state; a queued scalar load makes a counted wait prove nothing."""
    import isa_lds_inflight as chk
    asm = '''
kern:
	ds_read_b128 v[0:3], v20
	ds_read_b128 v[4:7], v20 offset:1024
	ds_read_b128 v[8:11], v20 offset:2048
	s_waitcnt lgkmcnt(2)
	v_mfma_f32_16x16x32_bf16 v[12:15], v[0:3], v[16:19], v[12:15]
	v_mfma_f32_16x16x32_bf16 v[12:15], v[4:7], v[16:19], v[12:15]
	s_waitcnt lgkmcnt(0)
	v_mfma_f32_16x16x32_bf16 v[12:15], v[8:11], v[16:19], v[12:15]
.LBB0_1:
	ds_read_b128 v[0:3], v20
	s_load_dwordx2 s[0:1], s[4:5], 0x0
	ds_read_b128 v[4:7], v20 offset:1024
	s_waitcnt lgkmcnt(1)
	v_mov_b32_e32 v30, v0
	s_endpgm
'''
    f = chk.findings(asm)
    assert [(ln - 1, l0 - 1) for _, ln, _, l0 in f] == [(7, 3), (15, 11)], f        # (line numbers of the stretch above, its first line is empty)


@pytest.mark.skipif(not os.path.isfile(HIPCC), reason='hipcc not available')
def test_halo3_counted_lds_waits_cover_every_operand():
    """conv_halo3.hip waits for its inline-asm LDS reads with hand-counted lgkmcnt values. Round 6 shipped a count that left the first use of a weight
    fragment two reads short (610 places in the ISA of the bf16 kernels): invisible to every single-process parity test -- an LDS read lands long
    before the use -- and NaNs in 3 of 4 runs once a second process shared the GPU. The counts are replayed over the compiler's output here, for the
    default build and for the early hand-over variant."""
    import isa_lds_inflight as chk

    def isa(flags):
        with tempfile.NamedTemporaryFile(suffix='.s') as f:
            subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-DMG_H3_BF16_ONLY'] + flags +
                           [os.path.join(ROOT, 'maggie_amd', 'csrc', 'conv_halo3.hip'), '-o', f.name], check=True, stderr=subprocess.DEVNULL, timeout=900)
            return open(f.name).read()

    with ThreadPoolExecutor(2) as ex:
        default, early = ex.map(isa, [[], ['-DMG_H3_EARLY=1']])
    for name, asm in (('default', default), ('early hand-over', early)):
        assert asm.count('ds_read_b128') > 1000, name                # the inline-asm reads are what is being checked
        f = chk.findings(asm)
        assert not f, (name, len(f), f[:3])
