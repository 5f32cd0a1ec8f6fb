# round 6: robustness of the final build (kernels.hip changed this round: pruned variants, claim_slot factored out, audit annotations on every launch) --
# determinism soak at two shapes, randomised differential stress (incl. the fused lookups), parity + full-shape suites with the sampler tile built as
# 256 and 2048 (variant libraries selected through $LEGION_LIB: the shipped library is not touched), and the server -> ipc_service soak.
#   make -C legion-1_amd/csrc variant VARIANT=ktile256 VARIANT_FLAGS=-DLEGION_KTILE=256; ... VARIANT=ktile2048 VARIANT_FLAGS=-DLEGION_KTILE=2048
#   bash profiles/r06_runs_robust.sh
O=gpurun_out/r06k
mkdir -p $O
python profiles/soak_determinism.py > $O/soak.log 2>&1; echo "soak rc=$?"; tail -n 3 $O/soak.log
LEGION_STRESS_N=150 LEGION_STRESS_CACHE_N=64 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomised > $O/pytest_stress.log 2>&1; echo "stress rc=$?"; tail -n 1 $O/pytest_stress.log
for KT in 256 2048; do
  LEGION_LIB=$GRAFT_REPO_ROOT/legion-1_amd/csrc/variants/liblegion_amd_ktile$KT.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_shape.py -m gpu -q -k "not synth_server" > $O/pytest_ktile_$KT.log 2>&1; echo "ktile $KT rc=$?"; tail -n 1 $O/pytest_ktile_$KT.log
done
LEGION_TEST_EPOCHS=60 timeout -k 10 600 python -m pytest tests/test_gpu_ipc.py -m gpu -q -k "server_binary_to_ipc_service or synth_dataset_source" > $O/pytest_soak_ipc.log 2>&1; echo "ipc soak rc=$?"; tail -n 1 $O/pytest_soak_ipc.log
LEGION_TEST_EPOCHS=3 LEGION_TEST_SERVED_EVERY=97 timeout -k 10 1000 python -m pytest tests/test_gpu_full_shape.py -q -k synth_server --durations=1 > $O/pytest_soak_served.log 2>&1; echo "served soak rc=$?"; tail -n 3 $O/pytest_soak_served.log
