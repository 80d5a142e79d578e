#!/bin/bash
# Timing-only variant of libtsg.so with extra compiler flags: tools/build_variant.sh NAME "-DFOO=1" -> ablate/libtsg_NAME.so
# (git-ignored; run with TSG_LIB_OVERRIDE=ablate/libtsg_NAME.so).
set -e
name=$1; flags=$2
src=ts-trajgen_amd/csrc
obj=/tmp/tsg_variant_$name
mkdir -p $obj ablate
for f in tsg_dense tsg_search tsg_gat tsg_gat_train tsg_rollout; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c $src/$f.hip -o $obj/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ablate/libtsg_$name.so $obj/*.o
ls -la ablate/libtsg_$name.so
