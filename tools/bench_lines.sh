# the round's bench lines once more on the frozen library, with roofline.traffic from profiles/pmc_traffic.json (gpurun -- bash tools/bench_lines.sh <tag>)
cd $GRAFT_REPO_ROOT
T=${1:-r05}
python bench.py > gpurun_out/${T}_a_bench.json 2> gpurun_out/${T}_a_bench.err
python bench.py --config silesia --steps 4 --warmup 1 > gpurun_out/${T}_m5_bench.json 2> gpurun_out/${T}_m5_bench.err
python bench.py --config mix5 --steps 4 --warmup 1 > gpurun_out/${T}_m2_bench.json 2> gpurun_out/${T}_m2_bench.err
python bench.py --split 127 --steps 4 --warmup 1 > gpurun_out/${T}_p127_bench.json 2> gpurun_out/${T}_p127_bench.err
python -c "
import json
for t in ('a','m5','m2','p127'):
    d=json.load(open('gpurun_out/${T}_%s_bench.json'%t)); r=d['roofline']; print(t, d['value'], r['frac'], r['traffic'], (d.get('cpu_baseline') or {}).get('value'))
"
