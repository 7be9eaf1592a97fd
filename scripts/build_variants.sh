#!/bin/bash
# Build A/B variants of libmau_hip.so: scripts/build_variants.sh <file.hip> name1:"-DFLAG=1 ..." name2:"..."
# -> metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_<name>.so  (only <file.hip> is recompiled per variant)
set -e
cd "$(dirname "$0")/../metadata-augmented-unet-for-lst-ndvi_amd/csrc"
make -j8 > /dev/null
SRC=$1; shift
mkdir -p ../variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-variable -Wno-unused-result"
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c $SRC -o ../variants/${SRC%.hip}_$name.o
    objs=""
    for f in capi conv3x3 conv3x3_bf16 conv3x3_wgrad_bf16 bn spatial head lstm ssim; do
      [ -f $f.hip ] || continue
      if [ "$f.hip" == "$SRC" ]; then objs="$objs ../variants/${SRC%.hip}_$name.o"; else objs="$objs $f.o"; fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../variants/libmau_$name.so
    echo built $name ) &
done
wait
