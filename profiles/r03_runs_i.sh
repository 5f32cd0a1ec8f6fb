# round-3: smoke matrix of bench.py's other modes after this round's changes (one MI355X); one line per mode
mkdir -p gpurun_out/r03i
run() { name=$1; shift; python bench.py "$@" --headline-only --cpu-baseline-seconds 0 --min-time 0.5 > gpurun_out/r03i/$name.json 2> gpurun_out/r03i/$name.err; echo "$name rc=$?"; python3 - <<PY
import json
try:
    d = json.loads([l for l in open('gpurun_out/r03i/$name.json') if l.startswith('{')][-1]); r = d.get('roofline') or {}
    print('  %-22s %.4f ms/batch  %.2f G edges/s  gather %s us frac %s  cache %s' % ('$name', d['ms_per_step'], d['value'] / 1e9, r.get('avg_launch_us'), r.get('frac'), json.dumps(d.get('cache'))[:150]))
except Exception as ex:
    print('  $name: no line', ex)
PY
}
run unified25 --cache unified
run unified100 --cache unified --cache-frac 1.0
run unified_topo --cache unified --topo-frac 0.3
run lp --task lp --batch 7998
run uk_union_3hop --workload uk-union --fanout 25,10,5
run uk_union_2hop --workload uk-union --fanout 25,10
run level --gather level
run intra --pipeline intra
run overlap --pipeline overlap
run papers_2hop --fanout 25,10
