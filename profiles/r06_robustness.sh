#!/bin/bash
# Host-side sanitizer run of the product: liblegion_amd built with AddressSanitizer + UBSan on every HOST object
# (make -C legion-1_amd/csrc asan-host; device code untouched), selected through $LEGION_LIB, on the CPU-reachable subset of the suite in the
# BUILD CONTAINER (no GPU): the NULL / out-of-range argument walk over the C ABI, the host logic (shard pitch, client-open refusals, the
# synth spec, the server binary's meta_config parsing and refusals), the symbol table, and the slab / semaphore / mirror / poisoned-pipe protocol of ipc_env.cpp with a
# fake producer and two fake consumer processes.   bash profiles/r06_robustness.sh > profiles/r06_robustness.log 2>&1
set -u
cd "$(dirname "$0")/.."
make -C legion-1_amd/csrc -j8 asan-host > /dev/null || { echo "asan-host build failed"; exit 1; }
export LEGION_LIB=$PWD/legion-1_amd/csrc/asan/liblegion_amd_asan.so
export LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1:strict_string_checks=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
echo "== library under test: $LEGION_LIB"
echo "== sanitizer runtime:  $LD_PRELOAD"
python - <<'PY'
import legion1_amd.capi as K
L = K.lib()
maps = open("/proc/self/maps").read()
assert "liblegion_amd_asan.so" in maps and "libclang_rt.asan" in maps and "csrc/liblegion_amd.so" not in maps
print("== loaded:", K.lib_path(), "(the shipped liblegion_amd.so is NOT mapped);", L.legion_version().decode())
PY
rc=0
echo "== 0. control: the instrumentation is live (a deliberately short output buffer handed to legion_synth_spec must be caught)"
python - > /tmp/r06_asan_control.txt 2>&1 <<'PY'
import ctypes as C
import legion1_amd.capi as K
L = K.lib()
libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p
short = libc.malloc(16)                     # LegionSynthSpec is 152 bytes
L.legion_synth_spec(b"products", 1.0, C.c_void_p(short))
print("NOT CAUGHT")
PY
if grep -q "AddressSanitizer: heap-buffer-overflow" /tmp/r06_asan_control.txt && ! grep -q "NOT CAUGHT" /tmp/r06_asan_control.txt; then
  echo "   caught: $(grep -m1 'ERROR: AddressSanitizer' /tmp/r06_asan_control.txt)"; grep -m3 "legion_synth_spec\|synth.hip" /tmp/r06_asan_control.txt | sed 's/^/   /'
else echo "   CONTROL FAILED: the sanitizer did not report the overflow"; rc=1; fi
echo "== 1. tests/abi_null_args.py"; python tests/abi_null_args.py | tail -n 3 || rc=1
echo "== 2. tests/ipc_env_cpu.py (producer + 2 consumers + poisoned pipe + stale semaphores)"; python tests/ipc_env_cpu.py producer "rb$$_" | grep PRODUCER_OK || rc=1
export LEGION_SERVER_BIN=$PWD/legion-1_amd/csrc/asan/legion_asan      # the server binary itself under the sanitizers: argv + meta_config parsing and refusals
echo "== 3. pytest: host logic, symbol table, IPC env on the CPU, the slab against the reference's own shm helper (oracle/_ref)"
python -m pytest tests/test_host_logic.py tests/test_capi_symbols.py tests/test_ipc_env_cpu.py tests/test_ref_shm_compat.py -q -p no:cacheprovider 2>&1 | tail -n 6 || rc=1
echo "== apart from the control in step 0, 'ERROR: AddressSanitizer' / 'runtime error:' must not occur in this log"
echo "== exit status $rc"
exit $rc
