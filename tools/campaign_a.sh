# round-end measurements, part A (gpurun -- bash tools/campaign_a.sh): GPU tests, the four profiled bench configurations
cd $GRAFT_REPO_ROOT
timeout 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gputest_final.log 2>&1; tail -3 gpurun_out/gputest_final.log
bash tools/prof_round.sh r04_a > gpurun_out/r04_a.log 2>&1
bash tools/prof_cfg.sh r04_m5 silesia 2 > gpurun_out/r04_m5.log 2>&1
bash tools/prof_cfg.sh r04_m2 mix5 2 > gpurun_out/r04_m2.log 2>&1
bash tools/prof_cfg.sh r04_p127 enwik9 2 "--split 127" > gpurun_out/r04_p127.log 2>&1
python -c "
import json
for t in ('r04_a','r04_m5','r04_m2','r04_p127'):
    d=json.load(open('gpurun_out/%s/bench.json'%t)); print(t, d['value'], d['roofline']['frac'], (d.get('cpu_baseline') or {}).get('value'), d.get('bit_exact_vs_cpu_baseline', d.get('bit_exact_vs_reference')))
"
