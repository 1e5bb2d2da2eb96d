# usage: bash tools/build_variant.sh <name> <source.hip> [-DFLAGS...]  -> ad-yolo_amd/variants/lib_<name>.so = the in-tree objects with
# <source.hip> recompiled under the given flags (A/B libraries for ADYOLO_LIB; built here, they travel with the snapshot)
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; src=$2; shift 2
mkdir -p $R/ad-yolo_amd/variants /tmp/variant_$name
base=$(basename $src .hip)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function "$@" -c $R/ad-yolo_amd/csrc/$src -o /tmp/variant_$name/$base.o
objs=""
for o in $R/ad-yolo_amd/csrc/build/*.o; do
  if [ "$(basename $o)" = "$base.o" ]; then objs="$objs /tmp/variant_$name/$base.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ad-yolo_amd/variants/lib_$name.so $objs
echo built lib_$name.so
