#!/bin/bash
# Sanitizer builds of the library's THREADED HOST CODE: the staging threads of
# sk_hostio.cpp and the shard threads / parallel merge sort of sk_shard.cpp, driven by tools/sanitize/host_driver.cpp
# with the injected shard backend -- no device, no kernel launch.  Host-only compile of the host sources (HIP headers in
# host mode; the kernel launchers of sk_kernels.hip stay unresolved and are never called), clang's libomp with the
# archer tool so that ThreadSanitizer understands the OpenMP synchronisation.  Run in the build container, never on
# the GPU box:   tools/sanitize/run_host_sanitizers.sh [outdir]
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${1:-/tmp/sk_sanitize}
mkdir -p "$OUT"
LLVM=/opt/rocm/lib/llvm
CXX=$LLVM/bin/clang++
SRC=$R/spectralkernels.jl_amd/csrc
HOST_SOURCES="sk_api sk_hostio sk_shard sk_joint sk_rules sk_tables sk_k0"
COMMON="-x c++ -std=c++17 -D__HIP_PLATFORM_AMD__=1 -I/opt/rocm/include -fopenmp -g -O1 -fPIC -fno-omit-frame-pointer -Wno-unused-value"
status=0
for mode in thread address,undefined; do
  tag=${mode%%,*}
  B=$OUT/$tag
  mkdir -p "$B"
  echo "== -fsanitize=$mode"
  pids=""
  for f in $HOST_SOURCES; do
    $CXX $COMMON -fsanitize=$mode -c "$SRC/$f.cpp" -o "$B/$f.o" 2> "$B/$f.log" &
    pids="$pids $!"
  done
  $CXX $COMMON -fsanitize=$mode -c "$R/tools/sanitize/host_driver.cpp" -o "$B/host_driver.o" 2> "$B/host_driver.log" &
  pids="$pids $!"
  for p in $pids; do wait $p || { echo "compile failed (see $B/*.log)"; status=1; }; done
  objs=""
  for f in $HOST_SOURCES host_driver; do objs="$objs $B/$f.o"; done
  $LLVM/bin/clang++ -fsanitize=$mode -fopenmp -o "$B/host_driver" $objs -Wl,--unresolved-symbols=ignore-all \
      -L/opt/rocm/lib -lamdhip64 -lrocfft -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$LLVM/lib -ldl -lpthread 2> "$B/link.log" \
      || { echo "link failed (see $B/link.log)"; status=1; continue; }
  if [ "$tag" = thread ]; then
    env OMP_TOOL_LIBRARIES=$LLVM/lib/libarcher.so ARCHER_OPTIONS="verbose=0" TSAN_OPTIONS="halt_on_error=0 exitcode=66 ignore_noninstrumented_modules=1" \
        OMP_NUM_THREADS=8 "$B/host_driver" > "$B/run.out" 2> "$B/run.err"
  else
    env ASAN_OPTIONS="detect_leaks=0 exitcode=66" UBSAN_OPTIONS="print_stacktrace=1 halt_on_error=0" OMP_NUM_THREADS=8 \
        "$B/host_driver" > "$B/run.out" 2> "$B/run.err"
  fi
  rc=$?
  reports=$(grep -c "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error:" "$B/run.err")
  echo "   exit code $rc, sanitizer reports: $reports, $(cat "$B/run.out" | tail -1)"
  [ $rc -ne 0 ] && { status=1; head -60 "$B/run.err"; }
done
exit $status
