# round-end measurements, part A (gpurun -- bash tools/r06_campaign_a.sh): GPU tests, the four profiled bench configurations
cd $GRAFT_REPO_ROOT
timeout 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gputest_final.log 2>&1; tail -3 gpurun_out/gputest_final.log
bash tools/prof_round.sh r06_a > gpurun_out/r06_a.log 2>&1
bash tools/prof_cfg.sh r06_m5 silesia 2 > gpurun_out/r06_m5.log 2>&1
bash tools/prof_cfg.sh r06_m2 mix5 2 > gpurun_out/r06_m2.log 2>&1
bash tools/prof_cfg.sh r06_p127 enwik9 2 "--split 127" > gpurun_out/r06_p127.log 2>&1
python -c "
import json
for t in ('r06_a','r06_m5','r06_m2','r06_p127'):
    d=json.load(open('gpurun_out/%s/bench.json'%t)); print(t, d['value'], d['roofline']['frac'], (d.get('cpu_baseline') or {}).get('value'), d.get('bit_exact_vs_cpu_baseline', d.get('bit_exact_vs_reference')))
"
bash tools/prof_cfg.sh r06_p8 enwik9 2 "--split 8" > gpurun_out/r06_p8.log 2>&1
# the 954-stream point: one chunk round holds every task (tasks of ~1 MB): --steps 1 --warmup 0 throughout
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_p954; mkdir -p $O
python bench.py --split 954 --steps 1 --warmup 0 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --split 954 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof.err
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU"; do
  T=$(echo $SET | tr ' ' '_')
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmc_$T -o p -- python3 bench.py --split 954 --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_$T.json 2> $O/pmc_$T.err
done
tail -c 300 $O/bench.json
