cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python bench.py --workload tree --steps 2 --warmup 1 > gpurun_out/r04_tree_bench.json 2> gpurun_out/r04_tree_bench.err
python bench.py --workload tree --steps 1 --warmup 0 > gpurun_out/r04_tree_bench_cold.json 2> gpurun_out/r04_tree_bench_cold.err
timeout 300 python tools/gpu_dp4_timers.py 4 text > gpurun_out/r04_timers_m3_text.txt 2>&1
python tools/gpu_batch_probe.py > gpurun_out/r04_batch_probe.txt 2>&1
KIND=mix5 DICT_MIB=1024 timeout 300 python tools/gpu_timers.py 2 4 > gpurun_out/r04_timers_m2_mix5.txt 2>&1
KIND=text timeout 300 python tools/gpu_timers.py 2 4 > gpurun_out/r04_timers_m2_text.txt 2>&1
mkdir -p gpurun_out/r04_full
timeout 900 python tools/gpu_fullsize.py gpurun_out/r04_full > gpurun_out/r04_full_m3.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/r04_m5_full_pmc -o p -- python3 tools/gpu_fullsize_cfg.py silesia_m5_d256m gpurun_out/r04_full > gpurun_out/r04_full_silesia.log 2>&1
timeout 1200 python tools/gpu_fullsize_cfg.py mix5_m2_d1024m_task0 gpurun_out/r04_full > gpurun_out/r04_full_mix5.log 2>&1
python -c "
import json
d=json.load(open('gpurun_out/r04_tree_bench.json')); print('tree', d['value'], d['bit_exact_vs_reference'], d['cpu_baseline']['value'], d['last_step_stats_rank0'])
d=json.load(open('gpurun_out/r04_tree_bench_cold.json')); print('tree cold', d['value'], d['bit_exact_vs_reference'], d['last_step_stats_rank0'])
for f in ('fullsize','fullsize_silesia_m5_d256m','fullsize_mix5_m2_d1024m_task0'):
    d=json.load(open('gpurun_out/r04_full/%s.json'%f)); print(f, d['encode_seconds'], d['MBps'], d['sha256'][:16], d['reference']['sha256'][:16], d.get('bit_exact_vs_reference'))
"
