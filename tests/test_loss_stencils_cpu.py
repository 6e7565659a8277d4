"""Host check of the loss stencils' per-cell arithmetic (maggie_amd/csrc/loss_stencils.h, shared by the HIP kernels of csrc/losses.hip):
the batched-load ("inner") forms against the general walks on every inner cell, and the ring enumeration as a bijection. Compiled with
g++ from the very header the kernels include -- runs without a GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('g++') is None, reason='g++ not available')
def test_batched_loss_stencils_equal_the_general_walks(tmp_path):
    exe = str(tmp_path / 'loss_stencils_check')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-o', exe, os.path.join(ROOT, 'tests', 'csrc', 'loss_stencils_check.cpp')])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
