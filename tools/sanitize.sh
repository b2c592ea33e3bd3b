#!/bin/bash
# CPU-only sanitizer runs (AddressSanitizer + UBSan; the GPU pool offers none): the host rules / MCTS library under the
# search tests and a FEN fuzzer, and the checkpoint reader under a byte-mutation fuzzer.  Run in the build container.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); O=/tmp/kami_sanitize; mkdir -p $O
SAN="-O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer"
g++ $SAN -fPIC -shared -I$R/include -o $O/libkamisearch.so $R/kami_amd/host/search_api.cpp -L$R/kami_amd -lkamihip -Wl,-rpath,$R/kami_amd -lpthread
g++ $SAN -I$R/kami_amd/csrc -o $O/fuzz_archive $R/tools/fuzz_archive.cpp
export LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python - <<PY
import sys; sys.path.insert(0, "$R")
from kami_amd import search as S
S.LIB_PATH = "$O/libkamisearch.so"
import pytest
sys.exit(pytest.main(["$R/tests/test_search.py", "-x", "-q", "-p", "no:cacheprovider"]))
PY
for s in 1 2 3; do python $R/tools/fuzz_fen.py $O/libkamisearch.so $s 40000; done
unset LD_PRELOAD
for s in 1 2 3; do $O/fuzz_archive $R/tests/golden/ref_checkpoint_f30_c8_r1.pt $s 4000; done
echo "sanitize: clean"
