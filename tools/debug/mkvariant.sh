#!/bin/bash
# bash tools/debug/mkvariant.sh NAME "-DFLAG ..." [file.hip ...]: variants/NAME.so = the current library with the named sources
# (default conv_b3.hip) recompiled with extra flags; the other objects come from vitta_amd/csrc/.build as they are.
set -e
cd "$(dirname "$0")/../.."
python -m vitta_amd.build > /dev/null
NAME=$1; FLAGS=$2; shift 2 || true
FILES=${@:-conv_b3.hip}
mkdir -p variants /tmp/variant_$NAME
OBJS=""
for o in vitta_amd/csrc/.build/*.o; do
  stem=$(basename $o | cut -d. -f1)
  skip=0
  for f in $FILES; do [ "$stem" == "${f%.hip}" ] && skip=1; done
  [ $skip == 0 ] && OBJS="$OBJS $o"
done
for f in $FILES; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c vitta_amd/csrc/$f -o /tmp/variant_$NAME/${f%.hip}.o
  OBJS="$OBJS /tmp/variant_$NAME/${f%.hip}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o variants/$NAME.so
echo variants/$NAME.so
