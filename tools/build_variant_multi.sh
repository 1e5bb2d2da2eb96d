# usage: bash tools/build_variant_multi.sh <name> "<-DFLAGS ...>" <source.hip> [<source.hip> ...] -> ad-yolo_amd/variants/lib_<name>.so = the
# in-tree objects with the listed sources recompiled under the flags (several files at once: A/B libraries for ADYOLO_LIB)
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; flags=$2; shift 2
mkdir -p $R/ad-yolo_amd/variants /tmp/variant_$name
for src in "$@"; do
  base=$(basename $src .hip)
  extra=""; [ "$base" = "wino4" ] && extra="-DW4_BRING=9"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function $extra $flags -c $R/ad-yolo_amd/csrc/$src -o /tmp/variant_$name/$base.o &
done
wait
objs=""
for o in $R/ad-yolo_amd/csrc/build/*.o; do
  b=$(basename $o)
  if [ -f /tmp/variant_$name/$b ]; then objs="$objs /tmp/variant_$name/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ad-yolo_amd/variants/lib_$name.so $objs
echo built lib_$name.so
