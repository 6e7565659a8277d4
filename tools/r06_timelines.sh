#!/bin/bash
# The in-kernel timelines DESIGN section 12.1 quotes, on one lease -> gpurun_out/r06_h3_timelines.txt
# (variant libraries: tools/build_variant.sh dbg conv_igemm.hip "-DMG_HALO_TIMING"; h3dbg / h3e1 / h3e2 conv_halo3.hip "-DMG_H3_TIMING [-DMG_H3_EXP=1|2]")
out=gpurun_out/r06_h3_timelines.txt
V=maggie_amd/_variants
{
echo "## round-2 halo kernel (conv_igemm.hip, -DMG_HALO_TIMING): cycles since the workgroup's start, wave 0 of every 64th tile"
for s in "4 128 128 64" "4 256 256 32"; do echo "== N Cin Cout HW = $s"; MAGGIE_LIB_PATH=$V/lib_dbg.so MG_HALO3=0 python tools/halo_timeline.py $s 2>&1 | grep block | head -3; done
echo; echo "## halo3 (conv_halo3.hip, -DMG_H3_TIMING): consumer wave 0 of every 32nd tile; wall = 100 MHz wall clock relative to the first workgroup"
for c in "4 128 128 64 8,64,3" "4 128 128 64 8,32,4" "4 128 128 64 8,64,1" "4 256 256 32 8,32,4" "1 128 128 64 8,64,3" "4 512 512 16 4,32,4"; do echo "== N Cin Cout HW cfg = $c"; MAGGIE_LIB_PATH=$V/lib_h3dbg.so python tools/h3_timeline.py $c 2>&1 | grep work | head -3; done
echo; echo "## the same, nothing of the layer in any L2 (H3_COLD=1: operands rewritten and 256 MB of other traffic before every launch)"
for c in "4 128 128 64 8,64,3" "4 256 256 32 8,32,4"; do echo "== $c"; H3_COLD=1 MAGGIE_LIB_PATH=$V/lib_h3dbg.so python tools/h3_timeline.py $c 2>&1 | grep work | head -3; done
echo; echo "## experiment 1 (-DMG_H3_EXP=1): producers stop after the first ring fill -- the consumer walk alone (results wrong, timing only)"
for c in "4 256 256 32 8,32,4" "4 256 128 64 8,64,3"; do echo "== $c"; MAGGIE_LIB_PATH=$V/lib_h3e1.so python tools/h3_timeline.py $c 2>&1 | grep work | head -2; done
echo; echo "## experiment 2 (-DMG_H3_EXP=2): consumers only pass the barriers -- the LDS-DMA stream alone"
for c in "4 256 256 32 8,32,4" "4 256 128 64 8,64,3"; do echo "== $c"; MAGGIE_LIB_PATH=$V/lib_h3e2.so python tools/h3_timeline.py $c 2>&1 | grep work | head -2; done
} > $out 2>&1
wc -l $out
