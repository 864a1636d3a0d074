#!/bin/bash
# A/B builds: tools/build_variant.sh NAME "-DFLAG=1 ..." [file.hip ...]  -> adcensus_amd/lib/NAME/libadcensus_hip.so
# (recompiles the listed sources (default: k_aggregate.hip) with the extra flags, links them with the regular objects;
#  select at run time with ADC_HIP_LIB=adcensus_amd/lib/NAME/libadcensus_hip.so)
set -e
cd "$(dirname "$0")/../adcensus_amd/csrc"
NAME=$1; FLAGS=$2; shift 2 || true
SRCS=${@:-k_aggregate.hip}
make -s
mkdir -p ../build/$NAME ../lib/$NAME
OBJS=""
for f in capi k_cost k_arms k_aggregate k_scanline k_wta k_refine k_voting k_paper; do
  if echo " $SRCS " | grep -q " $f.hip "; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-value -Wno-unused-result $FLAGS -c $f.hip -o ../build/$NAME/$f.o
    OBJS="$OBJS ../build/$NAME/$f.o"
  else
    OBJS="$OBJS ../build/$f.o"
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -o ../lib/$NAME/libadcensus_hip.so
echo "built adcensus_amd/lib/$NAME/libadcensus_hip.so"
