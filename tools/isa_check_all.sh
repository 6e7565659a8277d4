#!/bin/bash
# The in-flight-LDS-read check (tools/isa_lds_inflight.py) over every source with hand-counted s_waitcnt lgkmcnt: hipcc -S of the device code, no GPU.
# conv_halo3.hip is also covered by tests/test_isa_load_chains_cpu.py (25 s); conv_igemm.hip takes ~2.5 min per dtype variant to compile, which is why it
# lives here and not in the CPU suite. usage: tools/isa_check_all.sh   -> one 'findings: n' line per file / variant; exit 1 if any n > 0
root=$(cd $(dirname $0)/.. && pwd)
tmp=$(mktemp -d /tmp/isa_chk.XXXX)
rc=0
run() {   # name, flags...
  name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only "$@" -o $tmp/$name.s 2>/dev/null || { echo "$name: compile failed"; return; }
  out=$(python $root/tools/isa_lds_inflight.py $tmp/$name.s | tail -1)
  echo "$name: $out ($(grep -c ASMSTART $tmp/$name.s) inline-asm statements)"
  [ "$out" = "findings: 0" ] && touch $tmp/$name.ok
}
run halo3_default $root/maggie_amd/csrc/conv_halo3.hip &
run halo3_early -DMG_H3_EARLY=1 $root/maggie_amd/csrc/conv_halo3.hip &
run igemm_bf16 -DMG_CONV_T=1 $root/maggie_amd/csrc/conv_igemm.hip &
run igemm_f16 -DMG_CONV_T=3 $root/maggie_amd/csrc/conv_igemm.hip &
wait
for n in halo3_default halo3_early igemm_bf16 igemm_f16; do [ -f $tmp/$n.ok ] || rc=1; done
rm -rf $tmp
exit $rc
