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
