#!/bin/bash
# A/B build of the library with extra -D flags for ONE source file: tools/build_variant.sh <name> <source.hip> "<flags>" -> maggie_amd/_variants/lib_<name>.so
# (use with MAGGIE_LIB_PATH=maggie_amd/_variants/lib_<name>.so; the other objects come from build/obj of the current default build)
name=$1; src=$2; flags=$3
root=$(cd $(dirname $0)/.. && pwd)
mkdir -p $root/build/obj_$name $root/maggie_amd/_variants
first=$(head -1 $root/maggie_amd/csrc/$src)
objs=""
pids=""
if [[ "$first" == "// build-variants:"* ]]; then
  spec=${first#// build-variants:}; var=$(echo $spec | cut -d= -f1 | tr -d ' '); vals=$(echo $spec | cut -d= -f2 | tr ',' ' ')
  for v in $vals; do
    o=$root/build/obj_$name/$src.${var}_$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $root/maggie_amd/csrc/$src -o $o -D$var=$v $flags 2> $o.log &
    pids="$pids $!"; objs="$objs $o"
  done
else
  o=$root/build/obj_$name/$src.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $root/maggie_amd/csrc/$src -o $o $flags 2> $o.log &
  pids="$pids $!"; objs="$objs $o"
fi
wait $pids
others=$(ls $root/build/obj/*.o | grep -v "/$src")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/maggie_amd/_variants/lib_$name.so $others $objs && echo built $root/maggie_amd/_variants/lib_$name.so
