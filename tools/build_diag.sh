#!/bin/bash
# Diagnostic library for tools/wide_stamps.py and tools/t128_stamps.py: layers_mfma.hip with in-kernel s_memtime stamps
# (-DKAMI_WIDE_DIAG) linked with the product's other objects into kami_amd/csrc/build/libkamihip_diag.so (git-ignored;
# travels to the GPU box).  Run after `python -m kami_amd.build`.  Select it with KAMI_AB_LIB=<path>.
set -e
cd "$(dirname "$0")/../kami_amd/csrc"
RT=$(python3 -c 'import os,torch;print(os.path.join(os.path.dirname(torch.__file__),"lib"))' 2>/dev/null || echo /opt/rocm/lib)
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-pass-failed -ffp-contract=fast -DKAMI_WIDE_DIAG -c layers_mfma.hip -o build/layers_mfma_diag.o
g++ -shared -o build/libkamihip_diag.so build/kh_api.o build/encode.o build/forward_simple.o build/tower_mfma.o build/tower8_mfma.o build/layers_mfma_diag.o build/train.o -L$RT -lamdhip64 -Wl,-rpath,$RT -lpthread
ls -la build/libkamihip_diag.so
