# round-end measurements, part B (gpurun -- bash tools/r06_campaign_b.sh): the many-task tree (+ its PMC passes), section timers,
# the batch probe, and the three full-size pins with this round's kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/prof_tree.sh r06_tree > gpurun_out/r06_tree.log 2>&1
timeout 300 python tools/gpu_dp4_timers.py 4 text > gpurun_out/r06_timers_m3_text.txt 2>&1
CSC_DEV_LIB=csc_amd/csrc/build/dev_fine/libcsc_mi355x_timers.so timeout 300 python tools/gpu_dp4_trace.py 3000 9000 > gpurun_out/r06_trace_after.txt 2>&1
timeout 600 python tools/gpu_dec_speed.py > gpurun_out/r06_dec_speed.txt 2>&1
KIND=silesia DICT_MIB=256 timeout 300 python tools/gpu_timers.py 5 4 > gpurun_out/r06_timers_m5_silesia.txt 2>&1
KIND=mix5 DICT_MIB=1024 timeout 300 python tools/gpu_timers.py 2 4 > gpurun_out/r06_timers_m2_mix5.txt 2>&1
KIND=text timeout 300 python tools/gpu_timers.py 2 4 > gpurun_out/r06_timers_m2_text.txt 2>&1
timeout 600 python tools/gpu_dec_timers.py > gpurun_out/r06_dec_timers.txt 2>&1
python tools/gpu_batch_probe.py > gpurun_out/r06_batch_probe.txt 2>&1
timeout 300 python tools/gpu_hp.py > gpurun_out/r06_hp_text.txt 2>&1
mkdir -p gpurun_out/r06_full
timeout 900 python tools/gpu_fullsize.py gpurun_out/r06_full > gpurun_out/r06_full_m3.log 2>&1
timeout 900 python tools/gpu_fullsize_cfg.py silesia_m5_d256m gpurun_out/r06_full > gpurun_out/r06_full_silesia.log 2>&1
timeout 1200 python tools/gpu_fullsize_cfg.py mix5_m2_d1024m_task0 gpurun_out/r06_full > gpurun_out/r06_full_mix5.log 2>&1
python -c "
import json
d=json.load(open('gpurun_out/r06_tree/bench.json')); print('tree', d['value'], d['bit_exact_vs_reference'], d['cpu_baseline']['value'], d['last_step_stats_rank0'])
d=json.load(open('gpurun_out/r06_tree/bench_cold.json')); print('tree cold', d['value'], d['bit_exact_vs_reference'], d['last_step_stats_rank0'])
for f in ('fullsize','fullsize_silesia_m5_d256m','fullsize_mix5_m2_d1024m_task0'):
    d=json.load(open('gpurun_out/r06_full/%s.json'%f)); print(f, d['encode_seconds'], d['MBps'], d['sha256'][:16], d['reference']['sha256'][:16], d.get('bit_exact_vs_reference'))
"
