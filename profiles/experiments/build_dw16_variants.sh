#!/bin/bash
# experimental builds of the library whose conv1-dW objects (conv_dw16_pair.hip, conv1_dw_gather.hip, conv_dw16.hip; or the
# translation units named in $UNITS) are compiled with
# extra -D flags; everything else is the ablation build's objects.  usage: [UNITS="a b"] build_dw16_variants.sh name "-DFLAG ..." [name flags ...]
# -> cartpoleplusplus_amd/lib/libcartpolepp_hip_<name>.so, loaded with CARTPOLEPP_ABLATION=<name>
set -e
cd "$(dirname "$0")/../../cartpoleplusplus_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DCPP_ABLATION"
OBJ=../lib/obj
UNITS=${UNITS:-conv_dw16_pair conv1_dw_gather conv_dw16}
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  T=/tmp/dw16var_$name; mkdir -p $T
  for f in $UNITS; do hipcc $FLAGS $defs -c $f.hip -o $T/$f.o & done; wait
  # the ablation library's object list, with the three objects above replaced
  objs=""
  for o in $(ls $OBJ/*.o); do
    b=$(basename $o .o)
    case $b in exact_*) continue;; esac      # (the exact-products library's objects)
    skip=0; for f in $UNITS; do if [ "$b" = "$f" ] || [ "$b" = "abl_$f" ]; then skip=1; fi; done
    [ $skip = 1 ] && continue
    # release objects that have an ablation twin are dropped in favour of the twin
    if [ "${b#abl_}" = "$b" ] && [ -f $OBJ/abl_$b.o ]; then continue; fi
    objs="$objs $o"
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libcartpolepp_hip_$name.so $objs $(for f in $UNITS; do echo $T/$f.o; done) -L/opt/rocm/lib -lrccl
  echo built $name
done
