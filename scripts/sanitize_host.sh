#!/bin/bash
# the host side of libhector_mpc_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer.
# Builds a host-instrumented copy of the library (device code not instrumented: -fno-gpu-sanitize), then runs the C/C++
# programs that drive the C ABI -- tests/src/host_api_sweep.c (every batched entry point incl. error paths, scratch reuse,
# safe pass, device group), examples/legacy_tick.cpp, examples/batched.c, examples/batched_multi.c -- against it.
# Leak checking is off (the HIP runtime keeps process-lifetime allocations); every other ASan/UBSan report is fatal.
# Run on a GPU box:  gpurun -- 'bash scripts/sanitize_host.sh > gpurun_out/sanitize.txt 2>&1'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/sanitize; mkdir -p $OUT
RT=$(find /opt/rocm/lib/llvm -name 'libclang_rt.asan-x86_64.so' | head -1)
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g"
# (every translation unit to an object first: handed to one hipcc command together with .hip sources, the objects would be parsed as HIP)
CF="--offload-arch=gfx950 -O1 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-pass-failed"
pids=""
for src in hmpc_capi hmpc_group; do
  hipcc $CF $SAN -fno-gpu-sanitize -c hector_simulation_amd/csrc/$src.hip -o $OUT/$src.o & pids="$pids $!"
done
for g in 0 1 2 3; do
  hipcc $CF -DHMPC_VARIANT_GROUP=$g -c hector_simulation_amd/csrc/hmpc_variants.hip -o $OUT/hmpc_variants_$g.o & pids="$pids $!"
done
for p in $pids; do wait $p || exit 2; done
hipcc --offload-arch=gfx950 -shared -fPIC $SAN -fno-gpu-sanitize -shared-libsan $OUT/hmpc_capi.o $OUT/hmpc_group.o $OUT/hmpc_variants_0.o $OUT/hmpc_variants_1.o \
  $OUT/hmpc_variants_2.o $OUT/hmpc_variants_3.o -ldl -o $OUT/libhector_mpc_hip.so || exit 2
rm -f $OUT/*.o
CLANG=$(dirname $(dirname "$RT"))/../../../bin/clang
[ -x "$CLANG" ] || CLANG=/opt/rocm/lib/llvm/bin/clang
fail=0
run() {  # name, compiler driver mode, source, args...
  local name=$1 lang=$2 src=$3; shift 3
  $CLANG $lang -O1 $SAN -shared-libsan -Iinclude $src -L$OUT -lhector_mpc_hip -lm -lstdc++ -Wl,-rpath,$PWD/$OUT -Wl,-rpath,$(dirname $RT) -o $OUT/$name || { echo "COMPILE FAILED $name"; fail=1; return; }
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 $OUT/$name "$@"
  local rc=$?
  echo "== $name $* -> exit $rc"
  [ $rc -eq 0 ] || fail=1
}
# (round 4: with twelve kernel variants in the code object the sanitizer runtime's own exit-time CHECK described below for the
#  RCCL case also fires for this program; same rule: its result line + no sanitizer report of ours = clean)
tolerant() {  # name, expected result line, compile/run arguments of run()...
  local name=$1 want=$2; shift 2
  local prev=$fail
  run $name "$@" > $OUT/$name.log 2>&1
  cat $OUT/$name.log | grep -v "^    #"
  if grep -q "$want" $OUT/$name.log && ! grep -E "ERROR: AddressSanitizer|runtime error:" $OUT/$name.log > /dev/null; then
    if grep -q "sanitizer_allocator_device.h" $OUT/$name.log; then
      echo "== $name: program completed with the expected result; the sanitizer runtime's own exit-time CHECK (above) is not counted"
      fail=$prev
    fi
  fi
}
tolerant host_api_sweep "host API sweep ok" "-x c -std=c11" tests/src/host_api_sweep.c
run legacy_tick "-x c++ -std=c++17" examples/legacy_tick.cpp
run batched "-x c -std=c11" examples/batched.c
run batched_multi "-x c -std=c11" examples/batched_multi.c 3 p2p
run batched_multi_3contact "-x c -std=c11" examples/batched_multi.c 4 p2p 3
run friction_sweep "-x c -std=c11" examples/friction_sweep.c
# The RCCL transport (group of one).  Since round 3 this build of ROCm's ASan runtime trips over one of ITS OWN internal
# checks while the process exits -- "sanitizer_allocator_device.h:125 CHECK failed: !dev_runtime_unloaded_", raised under
# __cxa_finalize -> libamdhip64 -> libhsa-runtime64 with no frame of this library -- whenever librccl is loaded next to the
# round-3 code object (ten kernel variants); bisected: round-2 hmpc_capi + round-3 hmpc_group is clean, round-3 hmpc_capi +
# round-2 hmpc_group is not, none of the round-3 host changes (version check, device restore, device-side safe pass) matters.
# The program itself has completed by then: its result line is checked instead of the exit code, and any OTHER sanitizer
# report still fails the run.
prev_fail=$fail
run batched_multi_rccl "-x c -std=c11" examples/batched_multi.c 1 > $OUT/rccl.log 2>&1
cat $OUT/rccl.log | grep -v "^    #"
if grep -q "rc 0, group of 1 (rccl), 0 of 1000 not ok, 0 gathered rows differ" $OUT/rccl.log && \
   ! grep -E "ERROR: AddressSanitizer|runtime error:" $OUT/rccl.log > /dev/null; then
  if grep -q "sanitizer_allocator_device.h" $OUT/rccl.log; then
    echo "== batched_multi_rccl: program completed with the expected result; the sanitizer runtime's own exit-time CHECK (above) is not counted"
    fail=$prev_fail
  fi
fi
echo "sanitize_host: $([ $fail -eq 0 ] && echo 'CLEAN (no AddressSanitizer / UBSan report in this library; every program produced its result)' || echo 'FAILED')"
exit $fail
