#!/bin/bash
# Experiment builds of libwaxhip.so that differ in batch.hip's compile-time switches only (same ABI; loaded with WAX_HIP_LIB=...).
#   bash tools/build_variants.sh "name:-DFLAG=1 -DOTHER=2" ...   ->  wax_amd/lib/exp/libwaxhip_<name>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
python -c "import sys; sys.path.insert(0, '$R'); from wax_amd import build as b; b.build()"
mkdir -p "$R/wax_amd/lib/exp"; cd "$R/wax_amd/lib"
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$R/include" $flags -c "$R/wax_amd/csrc/batch.hip" -o exp/batch_$name.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o exp/libwaxhip_$name.so obj/kernels.o obj/multiscan.o exp/batch_$name.o obj/filter.o obj/rrf.o obj/engine.o \
    && echo "built $name ($flags)" ) &
done
wait
rm -f exp/*.o
