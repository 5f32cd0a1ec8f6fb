# HISTORICAL RECORD (rounds 3 / 4): this script rebuilds or REPLACES legion-1_amd/csrc/liblegion_amd.so in place -- a killed run leaves an invalid
# library behind.  Since round 5 a variant library is built with `make -C legion-1_amd/csrc variant VARIANT=... VARIANT_FLAGS=...` and selected through
# $LEGION_LIB (profiles/ab_kernels.sh, profiles/r05_runs_robust.sh); the shipped library is never touched.  Kept as the record of what was run.
[ "${LEGION_RUN_HISTORICAL:-0}" = 1 ] || { echo "$0: historical script that overwrites the shipped library; see its header (LEGION_RUN_HISTORICAL=1 to run it anyway)"; exit 1; }
# round 4, call g: the host mirror of the counters in the hand-off slab -- IPC end-to-end tests, then the server loop as a trainer sees it with
# the mirror (new) and with the library of the commit before ("old": trainer reads the counters with a blocking device copy), alternating
O=$GRAFT_REPO_ROOT/gpurun_out/r04p
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_ipc.py tests/test_gpu_parity.py -x -q -m gpu -k "ipc or poison or server or trainer or handoff" > $O/pytest_ipc.log 2>&1 || { tail -30 $O/pytest_ipc.log; exit 1; }
tail -2 $O/pytest_ipc.log
LIB=$GRAFT_REPO_ROOT/legion-1_amd/csrc/liblegion_amd.so
cp $LIB $O/lib_new.so
for round in 1 2; do
  for which in new old; do
    if [ $which = old ]; then cp profiles/ab/liblegion_amd_old.so $LIB; else cp $O/lib_new.so $LIB; fi
    for fan in 25,10 25,10,5; do
      timeout -k 10 300 python3 examples/serve_bench.py --scale 1.0 --fanout $fan --variants 'reference loop,pipelined' > $O/serve_${which}${round}_$fan.log 2>&1
      echo "$which #$round {$fan}: $(grep -E 'pipelined|reference' $O/serve_${which}${round}_$fan.log | tr -s ' ' | tr '\n' ';')"
    done
  done
done
cp $O/lib_new.so $LIB
