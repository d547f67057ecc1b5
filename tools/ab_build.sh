#!/bin/bash
# tools/ab_build.sh NAME "-DFLAG ..." file.hip [file.hip ...] -- a second build of librefign_hip.so with extra flags on the named sources
# (every other object is taken from the main build): refign_amd/lib/ab/librefign_hip_NAME.so, selected at run time with RFN_LIB=<path>.
set -e
name=$1; flags=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd)
make -C $R/refign_amd/csrc > /dev/null
O=$R/refign_amd/lib/ab/obj_$name; mkdir -p $O
objs=""
for f in $R/refign_amd/lib/obj/*.o; do
  b=$(basename $f .o); use=$f
  for s in "$@"; do if [ "$b" = "$(basename $s .hip)" ]; then
    extra=""; case $b in attn|f8|mfma_gemm) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $extra $flags -c $R/refign_amd/csrc/$b.hip -o $O/$b.o
    use=$O/$b.o
  fi; done
  objs="$objs $use"
done
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared $objs -o $R/refign_amd/lib/ab/librefign_hip_$name.so
echo $R/refign_amd/lib/ab/librefign_hip_$name.so
