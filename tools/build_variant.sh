#!/bin/bash
# tools/build_variant.sh NAME FILE.hip[,FILE2.hip...] "-DFLAG=1 ..." : kernel-experiment build -- recompiles the named translation
# units of csrc/ with extra flags and links them with the shipped objects into buffer-x_amd/csrc/variants/libbufferx_NAME.so (select
# it with BX_HIP_SO=...; *.so is git-ignored but travels to the GPU box).  The shipped library is untouched.
set -e
NAME=$1; SRCS=$2; FLAGS=$3
CS=$(cd "$(dirname "$0")/../buffer-x_amd/csrc" && pwd)
make -C "$CS" -j8 > /dev/null
mkdir -p "$CS/variants/_obj_$NAME"
OTHERS=$(ls "$CS"/_obj/*.o)
OBJS=""
for SRC in $(echo "$SRCS" | tr ',' ' '); do
  OBJ="$CS/variants/_obj_$NAME/${SRC%.hip}.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I"$CS/../../include" $FLAGS -c "$CS/$SRC" -o "$OBJ" &
  OBJS="$OBJS $OBJ"
  OTHERS=$(echo "$OTHERS" | grep -v "/${SRC%.hip}.o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS $OBJS -pthread -o "$CS/variants/libbufferx_$NAME.so"
echo "built $CS/variants/libbufferx_$NAME.so"
