#!/bin/bash
# Build an experiment variant of the library next to the product one:  bash tools/build_variant.sh <name> [-DFLAG=V ...]
# -> tools/variants/libbiu_<name>.so (git-ignored; travels to the GPU box).  Use with BIU_LIB_PATH=tools/variants/libbiu_<name>.so.
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/tools/variants/$NAME
mkdir -p $O
pids=()
for s in $R/bio_image_unet_amd/csrc/*.hip; do
  b=$(basename $s .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -I $R/include -I $R/bio_image_unet_amd/csrc "$@" -c $s -o $O/$b.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/variants/libbiu_$NAME.so $O/*.o
rm -rf $O
echo built $R/tools/variants/libbiu_$NAME.so
