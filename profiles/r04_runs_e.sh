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
