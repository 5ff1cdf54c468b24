#!/bin/bash
# Host-side AddressSanitizer + UBSan pass over the library's C files (ViT_hip.c, vit_gather_rccl.c, vit_config.c,
# Network_posix.c, vit_report.c): a second copy of the library with those five objects instrumented (the HIP objects are the
# ordinary ones -- GPU sanitizers are not available on this pool), loaded by the tests through $VIT_HIP_LIB under an ASan
# preload.  Usage:  tools/host_asan.sh build            (container or GPU box; needs the ordinary build's *.o)
#                   tools/host_asan.sh build-all        (also the host halves of the nine .hip files; a few minutes)
#                   tools/host_asan.sh test [pytest args]   e.g.  test -m "not gpu"    |    test -m gpu
# Three torch-interop tests cannot initialise torch under the preload (its own dlopen of libcaffe2_nvrtc.so fails); every
# other test runs.  Round 4: 34 / 34 CPU tests, 221 / 224 GPU tests, no report from either sanitizer.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/tools/scratch/asan"
CSRC="$ROOT/vit-with-opencl_amd/csrc"
case "${1:-}" in
build)
    mkdir -p "$OUT"
    for f in ViT_hip vit_gather_rccl vit_config Network_posix vit_report; do
        gcc -O1 -g -fPIC -ffp-contract=off -std=c11 -Wall -D_POSIX_C_SOURCE=200809L -fsanitize=address,undefined \
            -fno-omit-frame-pointer -I"$ROOT/include" -I"$CSRC" -c "$CSRC/$f.c" -o "$OUT/$f.o"
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvit_hip_asan.so" \
        "$CSRC"/{kernelHandler,gemm_mfma,gemm_p3,gemm_mx,attention_f32,attention_p3,attention_tiled,attention_h16,rowops}.o \
        "$OUT"/*.o -Wl,-rpath,/opt/rocm/lib -lm -lpthread -ldl -fsanitize=address,undefined -fno-gpu-sanitize
    echo "built $OUT/libvit_hip_asan.so" ;;
build-all)   # the HOST halves of the .hip files (launchers: argument checks, tile rules) instrumented as well; device code untouched
    mkdir -p "$OUT"
    for f in ViT_hip vit_gather_rccl vit_config Network_posix vit_report; do
        gcc -O1 -g -fPIC -ffp-contract=off -std=c11 -Wall -D_POSIX_C_SOURCE=200809L -fsanitize=address,undefined \
            -fno-omit-frame-pointer -I"$ROOT/include" -I"$CSRC" -c "$CSRC/$f.c" -o "$OUT/$f.o"
    done
    for f in kernelHandler gemm_mfma gemm_p3 gemm_mx attention_f32 attention_p3 attention_tiled attention_h16 rowops; do
        /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 -Wno-unused-function \
            -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
            -I"$ROOT/include" -I"$CSRC" -c "$CSRC/$f.hip" -o "$OUT/$f.hip.o" &
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvit_hip_asan.so" "$OUT"/*.o \
        -Wl,-rpath,/opt/rocm/lib -lm -lpthread -ldl -fsanitize=address,undefined -fno-gpu-sanitize
    echo "built $OUT/libvit_hip_asan.so (host halves of the .hip files instrumented too)" ;;
test)
    shift
    export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
    export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1
    export VIT_HIP_LIB="$OUT/libvit_hip_asan.so"
    cd "$ROOT" && python -m pytest tests -q "$@" ;;
*)  echo "usage: $0 build | build-all | test [pytest args]"; exit 2 ;;
esac
