#!/bin/bash
# The CPU suite with libgswt_host.so and the C oracle built under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: GPU sanitizers are not available on the pool).
# Works in a scratch clone so the product libraries stay as built.   usage: bash tools/host_sanitizers.sh [scratch dir]
set -eo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd); T=${1:-/tmp/gswt_san}
rm -rf "$T"; git clone -q "$ROOT" "$T"; cd "$T"
python -c "import __graft_entry__ as g; g.build()"
( cd gswt_renderer_amd/csrc && g++ -O1 -g -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fvisibility=hidden -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared host/gswt_host.cpp -o ../lib/libgswt_host.so -lz )
( cd oracle && F="-O1 -g -std=gnu11 -ffp-contract=off -fno-fast-math -fopenmp -fvisibility=hidden -shared -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer" && \
    gcc $F gswt_oracle.c -o _build/libgswt_oracle.so -lm && gcc $F -mfma gswt_oracle.c -o _build/libgswt_oracle_fma.so -lm )
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests -q -x -s -m "not gpu" -p no:cacheprovider 2>&1 | tee "$T/san.log" | tail -3
if grep -q "runtime error\|AddressSanitizer" "$T/san.log"; then echo "sanitizer findings: see $T/san.log"; exit 1; fi
echo "no sanitizer findings"
