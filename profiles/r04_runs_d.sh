# round 4, call d: (1) k_write with 4 prefix steps per round ("old" = the variant here) against the shipped kernels, same box; (2) the server loop
# variants at the full products shape, 20 epochs
O=$GRAFT_REPO_ROOT/gpurun_out/r04i
mkdir -p $O
bash $GRAFT_REPO_ROOT/profiles/ab_kernels.sh r04i/ab > $O/ab_write_prefix.log 2>&1; cat $O/ab_write_prefix.log
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 examples/serve_bench.py --scale 1.0 --fanout 25,10 > $O/serve_2hop.log 2>&1; cat $O/serve_2hop.log
timeout -k 10 500 python3 examples/serve_bench.py --scale 1.0 --fanout 25,10,5 > $O/serve_3hop.log 2>&1; cat $O/serve_3hop.log
