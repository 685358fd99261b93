#!/bin/bash
# CPU sanitizer run (CPU only - GPU AddressSanitizer is not available on this pool):
#   1. oracle/surfdisp_oracle.c with gcc -fsanitize=address,undefined      -> tests/test_oracle.py
#   2. tests/hostcheck (the group-velocity DEVICE math of surfdisp_kernels.hip compiled for the host) with
#      clang -fsanitize=address,undefined                                   -> tests/test_hostcheck.py
# Each under its own sanitizer runtime (LD_PRELOAD: the interpreter is not instrumented).  Log: profiles/<tag>/sanitizers_cpu.log
set -uo pipefail
cd "$(dirname "$0")/.."
TAG=${1:-r03a}
LOG=profiles/$TAG/sanitizers_cpu.log
mkdir -p profiles/$TAG
: > $LOG
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
echo "== 1. oracle: gcc $(gcc -dumpversion) -fsanitize=address,undefined" | tee -a $LOG
make -C oracle asan >> $LOG 2>&1 || exit 1
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" SURFDISP_ORACLE_LIB=$PWD/oracle/libsurfdisp_oracle_asan.so \
  python -m pytest tests/test_oracle.py -x -q -p no:cacheprovider 2>&1 | tail -15 | tee -a $LOG
echo "== 2. hostcheck: hipcc (clang) -fsanitize=address,undefined -fno-gpu-sanitize (host code instrumented, device code not)" | tee -a $LOG
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
$HIPCC -O1 -g -std=c++17 --offload-arch=gfx950 -fno-gpu-sanitize -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared-libsan \
  -Iinclude -Ipysurfinv_amd/csrc -shared -o tests/hostcheck/libhostcheck_asan.so tests/hostcheck/hostcheck.hip >> $LOG 2>&1 || { echo "hostcheck sanitizer build failed" | tee -a $LOG; exit 1; }
LD_PRELOAD="$RT" SURFDISP_HOSTCHECK_LIB=$PWD/tests/hostcheck/libhostcheck_asan.so \
  python -m pytest tests/test_hostcheck.py -x -q -p no:cacheprovider 2>&1 | tail -15 | tee -a $LOG
grep -c "ERROR: AddressSanitizer\|runtime error:" $LOG | sed 's/^/sanitizer reports in the log: /' | tee -a $LOG
