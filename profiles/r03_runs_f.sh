# HISTORICAL RECORD (rounds 3 / 4): this script rebuilds or REPLACES legion-1_amd/csrc/liblegion_amd.so in place -- a killed run leaves an invalid
# library behind.  Since round 5 a variant library is built with `make -C legion-1_amd/csrc variant VARIANT=... VARIANT_FLAGS=...` and selected through
# $LEGION_LIB (profiles/ab_kernels.sh, profiles/r05_runs_robust.sh); the shipped library is never touched.  Kept as the record of what was run.
[ "${LEGION_RUN_HISTORICAL:-0}" = 1 ] || { echo "$0: historical script that overwrites the shipped library; see its header (LEGION_RUN_HISTORICAL=1 to run it anyway)"; exit 1; }
# round-3 evidence, run F: the GPU suite after the last library change, the parity suite with the sampler tile built as 256 and 2048,
# the determinism soak at both shapes
mkdir -p gpurun_out/r03f
python -m pytest tests -m gpu -q > gpurun_out/r03f/pytest_default.log 2>&1; echo "default rc=$?"
python profiles/soak_determinism.py > gpurun_out/r03f/soak.log 2>&1; echo "soak rc=$?"; tail -3 gpurun_out/r03f/soak.log
for KT in 256 2048; do
  make -C legion-1_amd/csrc clean > /dev/null
  make -C legion-1_amd/csrc -j16 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result --offload-arch=gfx950 -DLEGION_KTILE=$KT" liblegion_amd.so legion > gpurun_out/r03f/build_$KT.log 2>&1; echo "build $KT rc=$?"
  python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_shape.py -m gpu -q > gpurun_out/r03f/pytest_ktile_$KT.log 2>&1; echo "ktile $KT rc=$?"; tail -1 gpurun_out/r03f/pytest_ktile_$KT.log
done
