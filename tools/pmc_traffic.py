"""Aggregate rocprofv3 --pmc counter_collection CSVs (FETCH_SIZE pass, WRITE_SIZE pass) per kernel family.
usage: pmc_traffic.py fetch.csv write.csv n_steps_profiled out.json
Units/corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B by the
gfx94x derived-counter formulas; on gfx950 FETCH_SIZE counts wide coalesced reads at HALF their bytes, so fetch bytes are
doubled. WRITE_SIZE is uncalibrated (reported as is)."""
import csv, json, sys, collections


def family(name):
    # the sparse head's launches (persistent row-count-from-device forms: every m_dev launch of the step; gather9 and the MODE = 2 per-tap weight
    # gradients) are kept apart from the dense trunk's (the head's own HBM bytes could not be read)
    if 'igemm_fprop' in name and 'persistent' in name: return 'igemm_fprop_sparse_head'
    if 'igemm_wgrad_gather9' in name: return 'igemm_wgrad_sparse_head'
    if 'igemm_wgrad_kernel' in name:
        args = name.split('igemm_wgrad_kernel<')[1].split('>')[0].split(',')
        if len(args) >= 4 and args[3].strip() == '2': return 'igemm_wgrad_sparse_head'
    if 'igemm_fprop' in name or 'conv_halo3' in name: return 'igemm_fprop'
    if 'igemm_wgrad' in name: return 'igemm_wgrad'
    if 'wgrad_reduce' in name: return 'wgrad_reduce'
    if 'anonymous namespace' in name and 'at::' not in name:
        return name.split('::')[1].split('<')[0].split('(')[0]
    return None


def load(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        f = family(r['Kernel_Name'])
        if f:
            acc[f][0] += float(r['Counter_Value']); acc[f][1] += 1
    return acc


fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
steps = float(sys.argv[3])
out = {}
for f in sorted(set(fetch) | set(write)):
    fs, fn = fetch.get(f, [0.0, 0]); ws, wn = write.get(f, [0.0, 0])
    n = max(fn, wn, 1)
    out[f] = {'launches_per_step': round(n / steps, 1),
              'fetch_bytes_per_launch_corrected': round(2.0 * 1024.0 * fs / max(fn, 1)),
              'write_bytes_per_launch': round(1024.0 * ws / max(wn, 1)),
              'hbm_bytes_per_launch': round(2.0 * 1024.0 * fs / max(fn, 1) + 1024.0 * ws / max(wn, 1)),
              'hbm_MB_per_step': round((2.0 * 1024.0 * fs + 1024.0 * ws) / steps / 1e6, 2)}
import subprocess
try:
    out['commit'] = subprocess.check_output(['git', '-C', __file__.rsplit('/', 2)[0], 'rev-parse', '--short', 'HEAD'], stderr=subprocess.DEVNULL).decode().strip()
except Exception:
    out['commit'] = None                                     # (no .git on the GPU box: the caller stamps it)
json.dump(out, open(sys.argv[4], 'w'), indent=1)
for f, d in sorted(((k, v) for k, v in out.items() if isinstance(v, dict)), key=lambda kv: -kv[1]['hbm_MB_per_step'])[:16]:
    print('%-28s %6.1f launches/step  %10.0f B/launch  %8.2f MB/step' % (f, d['launches_per_step'], d['hbm_bytes_per_launch'], d['hbm_MB_per_step']))
