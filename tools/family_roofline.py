"""Per-family roofline of one replayed step: kernel time (rocprofv3 --kernel-trace summary) x HBM-side bytes (separate --pmc FETCH_SIZE / WRITE_SIZE passes,
tools/pmc_traffic.py) -> achieved GB/s against the 8 TB/s HBM3E peak, for every family with at least 0.03 ms per step.
usage: python tools/family_roofline.py profiles/r05_trace_summary.txt profiles/r05_pmc_traffic.json > profiles/r05_family_roofline.txt"""
import json, re, sys
tr, pm = open(sys.argv[1]).read().splitlines(), json.load(open(sys.argv[2]))
PEAK = 8000.0
fam = {}
for l in tr[1:]:
    m = re.match(r'\s+(?:maggie: |other: )?(.+?)\s+([\d.]+) launches/step\s+([\d.]+) ms/step\s+([\d.]+) us avg', l)
    if m:
        fam[m.group(1).strip()] = (float(m.group(2)), float(m.group(3)))


def mb_of(name):
    """HBM-side MB per step of a trace family: the PMC table is keyed by kernel name; the conv families share one key each (igemm_fprop / igemm_wgrad)."""
    key = name.split(' ')[0].split('(')[0]
    if key == 'void':
        key = name.split(' ')[1].split('<')[0]
    v = pm.get(key)
    return v.get('hbm_MB_per_step') if isinstance(v, dict) else None


print(tr[0])
print('%-52s %9s %9s %11s %9s %7s' % ('family', 'launches', 'ms/step', 'HBM MB/step', 'GB/s', '% of 8T'))
for name, (n, ms) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    if ms < 0.03:
        continue
    mb = None if name.startswith(('igemm_', 'conv_halo3')) else mb_of(name)
    if mb:
        gbs = mb / ms
        print('%-52s %9.1f %9.3f %11.1f %9.0f %6.1f%%' % (name[:52], n, ms, mb, gbs, 100.0 * gbs / PEAK))
    else:
        print('%-52s %9.1f %9.3f %11s %9s %7s' % (name[:52], n, ms, '-', '-', '-'))

# the two conv families as a whole (the PMC pass runs eagerly and keys all forms of a family by one name)
for key, pat in (('igemm_fprop', 'igemm_fprop'), ('igemm_wgrad', 'igemm_wgrad')):
    match = lambda k: k.startswith(pat) or (key == 'igemm_fprop' and k.startswith('conv_halo3'))
    ms = sum(v[1] for k, v in fam.items() if match(k)) + (fam.get('splitk_finish_kernel', (0, 0))[1] if key == 'igemm_fprop' else 0.0)
    n = sum(v[0] for k, v in fam.items() if match(k))
    v = pm.get(key)
    if isinstance(v, dict) and ms:
        mb = v['hbm_MB_per_step']
        print('%-52s %9.1f %9.3f %11.1f %9.0f %6.1f%%   (MFMA-bound family: see roofline.frac)' % (key + ' family, all forms (incl. sparse head)', n, ms, mb, mb / ms, 100.0 * mb / ms / PEAK))
