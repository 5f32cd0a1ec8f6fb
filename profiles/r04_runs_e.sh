# HISTORICAL RECORD (rounds 3 / 4): this script rebuilds or REPLACES legion-1_amd/csrc/liblegion_amd.so in place -- a killed run leaves an invalid
# library behind.  Since round 5 a variant library is built with `make -C legion-1_amd/csrc variant VARIANT=... VARIANT_FLAGS=...` and selected through
# $LEGION_LIB (profiles/ab_kernels.sh, profiles/r05_runs_robust.sh); the shipped library is never touched.  Kept as the record of what was run.
[ "${LEGION_RUN_HISTORICAL:-0}" = 1 ] || { echo "$0: historical script that overwrites the shipped library; see its header (LEGION_RUN_HISTORICAL=1 to run it anyway)"; exit 1; }
# round 4, call e: robustness of the final build -- determinism soak at two shapes, randomised differential stress (incl. the fused lookups),
# parity + full-shape suites with the sampler tile built as 256 and 2048 (the full-shape oracle comparison included)
O=gpurun_out/r04k
mkdir -p $O
python profiles/soak_determinism.py > $O/soak.log 2>&1; echo "soak rc=$?"; tail -3 $O/soak.log
LEGION_STRESS_N=150 LEGION_STRESS_CACHE_N=64 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomised > $O/pytest_stress.log 2>&1; echo "stress rc=$?"; tail -1 $O/pytest_stress.log
cp legion-1_amd/csrc/liblegion_amd.so $O/lib_default.so
for KT in 256 2048; do
  make -C legion-1_amd/csrc clean > /dev/null
  make -C legion-1_amd/csrc -j16 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result --offload-arch=gfx950 -DLEGION_KTILE=$KT" liblegion_amd.so legion > $O/build_$KT.log 2>&1; echo "build $KT rc=$?"
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_shape.py -m gpu -q > $O/pytest_ktile_$KT.log 2>&1; echo "ktile $KT rc=$?"; tail -1 $O/pytest_ktile_$KT.log
done
