#!/bin/bash
# Host-side sanitizer pass (CPU only; GPU sanitizers are not available on this pool): the CPU oracle and the host-only GGUF
# parser of the library, built with AddressSanitizer + UBSan, under the oracle's KAT / golden / generator tests and a
# 3000-case mutation fuzz of the parser.  Usage: bash tools/sanitize/run.sh   (from the repo root; writes only under /tmp)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
SAN="-O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer"
ASAN_LIB=$(g++ -print-file-name=libasan.so)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:allocator_may_return_null=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
g++ $SAN -ffp-contract=off -fno-fast-math -pthread -o /tmp/liboracle_asan.so "$R"/oracle/quant.cpp "$R"/oracle/ops.cpp "$R"/oracle/model.cpp "$R"/oracle/turboquant.cpp
g++ $SAN -I"$R"/include -o /tmp/libgguf_asan.so "$R"/llama-gguf_amd/csrc/gguf_loader.cpp "$R"/tools/sanitize/gguf_stubs.cpp
cp "$R"/oracle/liboracle.so /tmp/liboracle_real.so
trap 'cp /tmp/liboracle_real.so "$R"/oracle/liboracle.so; touch "$R"/oracle/liboracle.so' EXIT
cp /tmp/liboracle_asan.so "$R"/oracle/liboracle.so; touch "$R"/oracle/liboracle.so
(cd "$R" && LD_PRELOAD=$ASAN_LIB python -m pytest tests/test_oracle_kat.py tests/test_golden.py tests/test_synth.py -q -x -m "not gpu")
(cd "$R" && LD_PRELOAD=$ASAN_LIB LGH_GGUF_LIB=/tmp/libgguf_asan.so python tools/sanitize/gguf_fuzz.py)
