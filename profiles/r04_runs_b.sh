# round 4, call b: (1) the server's RunOnce variants traced (plain two-stream loop, graph modes 1 / 2 / 3), (2) L2 / EA request counters and
# (3) HBM byte counters at the products {25,10,5} shape, (4) the random-access ceilings at the products table sizes
O=$GRAFT_REPO_ROOT/gpurun_out/r04g
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_ipc.py -x -q -m gpu -k "server_binary" > $O/pytest_ipc.log 2>&1 || { tail -30 $O/pytest_ipc.log; exit 1; }
tail -2 $O/pytest_ipc.log
timeout -k 10 500 python3 profiles/graph_trace.py --fanout 25,10 --out $O/gt2 > $O/graph_trace_2hop.log 2>&1; tail -30 $O/graph_trace_2hop.log
timeout -k 10 500 python3 profiles/graph_trace.py --fanout 25,10,5 --out $O/gt3 --epochs 4 > $O/graph_trace_3hop.log 2>&1; tail -30 $O/graph_trace_3hop.log
python3 profiles/pmc_sq.py l2 --workload products --fanout 25,10,5 > $O/pmc_l2_products_3hop.log 2>&1; tail -14 $O/pmc_l2_products_3hop.log
python3 profiles/make_pmc_traffic.py r04 products_3hop --workload products --fanout 25,10,5 > $O/pmc_traffic_products_3hop.log 2>&1; tail -2 $O/pmc_traffic_products_3hop.log
python3 profiles/probe_rate.py > $O/probe_rate.log 2>&1; tail -12 $O/probe_rate.log
