# r06: per-workgroup LDS hash dedup in k_sample (-DLEGION_TILE_HASH) -- parity through the variant library, then a same-box A/B against the shipped one.
#   make -C legion-1_amd/csrc variant VARIANT=tilehash VARIANT_FLAGS=-DLEGION_TILE_HASH      (build container)
#   gpurun -- bash profiles/r06_runs_tilehash.sh
O=gpurun_out/r06_tilehash; mkdir -p $O
V=$GRAFT_REPO_ROOT/legion-1_amd/csrc/variants/liblegion_amd_tilehash.so
LEGION_LIB=$V python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_shape.py -m gpu -q -x -k "not server and not ipc" > $O/parity.log 2>&1; echo "variant parity rc=$?" | tee -a $O/parity.log
tail -3 $O/parity.log
bash profiles/ab_kernels.sh r06_tilehash/ab $V "papers100M 25,10,5" "products 25,10,5" "products 25,10" "uk-union 25,10" > $O/ab.log 2>&1; echo "ab rc=$?"
cat $O/ab.log
