"""MFMA utilisation per kernel family from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES pass.
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs)  (the gfx94x derived-counter formula)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
def fam(name):
    if 'igemm_fprop' in name or 'conv_halo3' in name: return 'igemm_fprop'
    if 'igemm_wgrad' in name: return 'igemm_wgrad'
    return None
for r in csv.DictReader(open(sys.argv[1])):
    f = fam(r['Kernel_Name'])
    if f:
        acc[f][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': n[f] += 1
NXCD = 8     # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs of an MI355X: the device's active cycles are value / 8
for f, c in acc.items():
    active = c['GRBM_GUI_ACTIVE'] / NXCD
    util = c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(active * 256 * 4, 1)
    print('%-12s launches %d  MFMA busy cycles %.3g  device active cycles %.3g (GRBM_GUI_ACTIVE / %d XCDs)  MfmaUtil %.2f %%'
          % (f, n[f], c['SQ_VALU_MFMA_BUSY_CYCLES'], active, NXCD, 100 * util))
