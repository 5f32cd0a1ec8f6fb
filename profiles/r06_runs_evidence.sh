# r06 evidence at the round's final code (GPU box, repository root):  bash profiles/r06_runs_evidence.sh
O=gpurun_out/r06_evidence; mkdir -p $O
python3 profiles/collect_round.py r06 > $O/collect.log 2>&1; echo "collect rc=$?"; tail -3 $O/collect.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_driver_cmd.json 2> $O/driver_cmd.err; echo "driver cmd rc=$?"
LEGION_DEVICE_AUDIT=1 LEGION_BENCH_FORCE_DEVICE=0 python3 bench.py --gpus 2 --scale 0.2 --steps 20 > $O/r06_bench_2rank_rehearsal.json 2> $O/rehearsal2.err; echo "2-rank rehearsal (audit on) rc=$?"
python3 -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
cp profiles/r06_bench_line.json profiles/r06_bench_papers100M_kernel_stats.csv profiles/r06_bench_papers100M_summary.md profiles/r06_pmc_hbm_traffic.json $O/ 2>/dev/null
