#!/bin/bash
# Build A/B variants of libmau_hip.so: scripts/build_variants.sh <file.hip> name1:"-DFLAG=1 ..." name2:"..."
# -> metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_<name>.so  (only <file.hip> is recompiled per variant)
set -e
cd "$(dirname "$0")/../metadata-augmented-unet-for-lst-ndvi_amd/csrc"
make -j8 > /dev/null
SRC=$1; shift
mkdir -p ../variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-variable -Wno-unused-result"
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( # the conv kernel's instantiations live in six translation units that include conv3x3_bf16.hip: a variant rebuilds all seven
    if [ "$SRC" == "conv3x3_bf16.hip" ]; then parts="conv3x3_bf16 conv3x3_bf16_e0_bf16 conv3x3_bf16_e1_bf16 conv3x3_bf16_e2_bf16 conv3x3_bf16_e0_f16 conv3x3_bf16_e1_f16 conv3x3_bf16_e2_f16"; else parts="${SRC%.hip}"; fi
    for part in $parts; do /opt/rocm/bin/hipcc $FLAGS $defs -c $part.hip -o ../variants/${part}_$name.o & done; wait
    objs=""
    for f in capi conv3x3 conv3x3_bf16 conv3x3_bf16_e0_bf16 conv3x3_bf16_e1_bf16 conv3x3_bf16_e2_bf16 conv3x3_bf16_e0_f16 conv3x3_bf16_e1_f16 conv3x3_bf16_e2_f16 conv3x3_wgrad_bf16 conv3x3_wgrad16 conv3x3_first bn bn_fused spatial head lstm ssim optim embfold; do
      [ -f $f.hip ] || continue
      if [[ " $parts " == *" $f "* ]]; then objs="$objs ../variants/${f}_$name.o"; else objs="$objs $f.o"; fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../variants/libmau_$name.so
    echo built $name )
done
