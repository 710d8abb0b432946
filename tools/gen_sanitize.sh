#!/bin/bash
# the generator's host code under ASan + UBSan (CPU only; GPU sanitizers are not available): see tools/gen_sanitize.py
set -e
cd "$(dirname "$0")/.."
make -C halo2-gpu-specific_amd/csrc -j8 > /dev/null
S=$(mktemp -d /tmp/h2_san_XXXXXX)
CS=halo2-gpu-specific_amd/csrc
g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -std=c++17 -fPIC -I/opt/rocm/include \
    -Wa,-I$CS -c $CS/evalh_gen.cpp -o $S/evalh_gen.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $CS/context.o $CS/ntt.o $CS/poly.o $CS/msm.o $CS/evalh.o $CS/scan.o $CS/logup.o \
    $CS/capi.o $S/evalh_gen.o -ldl -o $S/libhalo2_hip.so
mkdir -p $S/cache && chmod 700 $S/cache
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
    UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 H2_LIB=$S/libhalo2_hip.so H2_JIT_CACHE=$S/cache python3 tools/gen_sanitize.py "$@"
rm -rf $S
