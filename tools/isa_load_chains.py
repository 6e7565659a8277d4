"""Find kernels whose loads are a dependent chain -- on the ISA, without a GPU.

usage: python tools/isa_load_chains.py [source.hip ...]      (default: every maggie_amd/csrc/*.hip except the conv family, which takes minutes to compile)

For each source: `hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only`, then per kernel the number of global / buffer loads, the number of FULL waits
(`s_waitcnt vmcnt(0)`) and the scratch size. A kernel with about as many full waits as loads issues one load per memory round trip (a run-time-length loop, or taps behind
run-time conditions): the candidates of DESIGN.md 11.13. Static counts -- loops are counted once -- so read the flagged kernels' ISA before acting on the numbers."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def kernels(asm):
    lines = asm.split('\n')
    i = 0
    while i < len(lines):
        m = re.match(r'^(_Z\w+):', lines[i])
        if m and '.type' in ''.join(lines[max(0, i - 6):i]):
            name, j, loads, w0, wn = m.group(1), i, 0, 0, 0
            while j < len(lines) and not lines[j].startswith('.Lfunc_end'):
                s = lines[j]
                loads += ('global_load' in s) or ('buffer_load' in s)
                w0 += 's_waitcnt vmcnt(0)' in s
                wn += 's_waitcnt vmcnt' in s
                j += 1
            scratch = 0
            for k in range(j, min(j + 80, len(lines))):
                mm = re.search(r'; ScratchSize: (\d+)', lines[k])
                if mm:
                    scratch = int(mm.group(1))
                    break
            yield name, loads, w0, wn, scratch
            i = j
        i += 1


def main():
    srcs = sys.argv[1:] or [s for s in sorted(glob.glob(os.path.join(ROOT, 'maggie_amd', 'csrc', '*.hip'))) if 'conv_' not in os.path.basename(s)]
    print('%-18s %5s %5s %5s %7s  kernel' % ('source', 'loads', 'full', 'waits', 'scratch'))
    for src in srcs:
        with tempfile.NamedTemporaryFile(suffix='.s') as f:
            subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', src, '-o', f.name], check=True, stderr=subprocess.DEVNULL)
            asm = open(f.name).read()
        for name, loads, w0, wn, scratch in kernels(asm):
            if scratch or (loads >= 6 and w0 >= 0.5 * loads):
                dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
                print('%-18s %5d %5d %5d %7d  %s' % (os.path.basename(src), loads, w0, wn, scratch, dem[:120]))


if __name__ == '__main__':
    main()
