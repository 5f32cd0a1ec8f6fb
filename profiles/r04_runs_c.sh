# round 4, call c: the server's RunOnce variants traced (plain two-stream loop, graph modes 1 / 2 / 3)
O=$GRAFT_REPO_ROOT/gpurun_out/r04h
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 profiles/graph_trace.py --fanout 25,10 --out $O/gt2 > $O/graph_trace_2hop.log 2>&1; tail -30 $O/graph_trace_2hop.log
timeout -k 10 500 python3 profiles/graph_trace.py --fanout 25,10,5 --out $O/gt3 > $O/graph_trace_3hop.log 2>&1; tail -30 $O/graph_trace_3hop.log
