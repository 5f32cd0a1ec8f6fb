# one full bench line (all legs of an N = 1 run, CPU baselines included) per remaining BASELINE config, round-3 code
mkdir -p gpurun_out/r03t
python bench.py --workload products --fanout 25,10,5 > gpurun_out/r03t/bench_products_3hop.json 2> gpurun_out/r03t/p3.err; echo "products 3-hop rc=$?"
python bench.py --workload uk-union --fanout 25,10 --no-cpu-features > gpurun_out/r03t/bench_uk_union_2hop.json 2> gpurun_out/r03t/uk2.err; echo "uk-union 2-hop rc=$?"
python bench.py --task lp --batch 7998 > gpurun_out/r03t/bench_lp.json 2> gpurun_out/r03t/lp.err; echo "lp rc=$?"
python bench.py > gpurun_out/r03t/bench_default.json 2> gpurun_out/r03t/default.err; echo "default rc=$?"
