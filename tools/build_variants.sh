#!/bin/bash
# Variant builds of the engine for same-device A/B runs (tools/tower_ablate.py, tools/encode_ab.py): the sources named in
# VARIANT_SRCS (default "tower_mfma tower8_mfma") compiled with extra -D flags, linked with the product's other objects
# into kami_amd/csrc/build/libkamihip_<name>.so (git-ignored; travels to the GPU box).  Run after `python -m kami_amd.build`.
#     tools/build_variants.sh name1:-DKAMI_TOWER_ABL=1 name2:"-DKAMI_TOWER_ABL=3 -DX=1" ...
set -e
cd "$(dirname "$0")/../kami_amd/csrc"
SRCS=${VARIANT_SRCS:-"tower_mfma tower8_mfma"}
ALL="kh_api encode forward_simple tower_mfma tower8_mfma layers_mfma train"
RT=$(python3 -c 'import os,torch;print(os.path.join(os.path.dirname(torch.__file__),"lib"))' 2>/dev/null || echo /opt/rocm/lib)
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  for src in $SRCS; do
    extra=""; case $src in tower*) extra="-mllvm -amdgpu-mfma-vgpr-form";; esac
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-pass-failed -Wno-unused-variable -ffp-contract=fast \
          $extra $flags -c $src.hip -o build/${src}_$name.o &
  done
done
wait
for spec in "$@"; do
  name="${spec%%:*}"
  objs=""
  for o in $ALL; do
    case " $SRCS " in *" $o "*) objs="$objs build/${o}_$name.o";; *) objs="$objs build/$o.o";; esac
  done
  g++ -shared -o build/libkamihip_$name.so $objs -L$RT -lamdhip64 -Wl,-rpath,$RT -lpthread
  ls -la build/libkamihip_$name.so
done
