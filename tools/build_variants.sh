#!/bin/bash
# Variant builds of the engine for same-device A/B runs (tools/tower_ablate.py): tower_mfma.hip / tower8_mfma.hip compiled with extra
# -D flags, linked with the product's other objects into kami_amd/csrc/build/libkamihip_<name>.so (git-ignored; travels
# to the GPU box).  Run after `python -m kami_amd.build`.
#     tools/build_variants.sh name1:-DKAMI_TOWER_ABL=1 name2:"-DKAMI_TOWER_ABL=3 -DX=1" ...
set -e
cd "$(dirname "$0")/../kami_amd/csrc"
RT=$(python3 -c 'import os,torch;print(os.path.join(os.path.dirname(torch.__file__),"lib"))' 2>/dev/null || echo /opt/rocm/lib)
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-pass-failed -Wno-unused-variable -ffp-contract=fast \
        -mllvm -amdgpu-mfma-vgpr-form $flags -c tower_mfma.hip -o build/tower_mfma_$name.o &
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-pass-failed -Wno-unused-variable -ffp-contract=fast \
        -mllvm -amdgpu-mfma-vgpr-form $flags -c tower8_mfma.hip -o build/tower8_mfma_$name.o &
done
wait
for spec in "$@"; do
  name="${spec%%:*}"
  g++ -shared -o build/libkamihip_$name.so build/kh_api.o build/encode.o build/forward_simple.o build/tower_mfma_$name.o build/tower8_mfma_$name.o build/layers_mfma.o build/train.o -L$RT -lamdhip64 -Wl,-rpath,$RT -lpthread
  ls -la build/libkamihip_$name.so
done
