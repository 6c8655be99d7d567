#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." file.hip [file.hip ...]: a measuring build of the TESTING library in which the named sources are
# compiled with the extra flags (every other object is the testing build's own) -> cniic_amd/libcniic_hip_NAME.so; run a tool against it with
# CNIIC_LIB_FILE=libcniic_hip_NAME.so.  Never shipped, never loaded by the tests.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; flags=$2; shift 2
cd $R/cniic_amd/csrc
make -s -j8 >/dev/null
objs=""
for o in *.t.o; do
  src=${o%.t.o}
  use=$o
  for f in "$@"; do
    if [ "$src.hip" == "$f" ] || [ "$src.cpp" == "$f" ]; then
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -DCNIIC_TESTING $flags -x hip -c $f -o /tmp/_variant_$name_$src.o
      use=/tmp/_variant_$name_$src.o
    fi
  done
  objs="$objs $use"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcniic_hip_$name.so $objs -ldl
echo built cniic_amd/libcniic_hip_$name.so
