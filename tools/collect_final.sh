# after `gpurun -- 'bash tools/campaign_a.sh; bash tools/campaign_b.sh'`: everything under gpurun_out/ that is judged -> profiles/
set -e
cd "$(dirname "$0")/.."
python tools/prof_collect.py r04_a | tail -4
python tools/prof_collect_cfg.py r04_m5 m5_d256m_single_stream "silesia stand-in -m5 -d256m, one stream, 2 chunks of 2 MiB" | tail -3
python tools/prof_collect_cfg.py r04_m2 m2_d1024m_single_stream "mix5 -m2 -d1024m, one stream, 2 chunks of 2 MiB" | tail -3
python tools/prof_collect_cfg.py r04_p127 m3_d64m_p127 "enwik9 stand-in -m3 -d64m as -p127 tasks, 2 chunk rounds of 127 x 2 MiB" | tail -3
cp gpurun_out/r04_tree_bench.json profiles/r04_tree_bench.json
cp gpurun_out/r04_tree_bench_cold.json profiles/r04_tree_bench_cold.json
for f in r04_timers_m3_text r04_timers_m2_mix5 r04_timers_m2_text r04_batch_probe; do cp gpurun_out/$f.txt profiles/$f.txt; done
sed -i 's/literal trees, eight at a time/waiting for the tree wavefront (d6_join)/' profiles/r04_timers_m3_text.txt
cp gpurun_out/r04_full/fullsize_silesia_m5_d256m.json profiles/r04_fullsize_silesia_m5_d256m.json
cp gpurun_out/r04_full/fullsize_mix5_m2_d1024m_task0.json profiles/r04_fullsize_mix5_m2_d1024m_task0.json
cp gpurun_out/r04_full/fullsize.json profiles/r04_fullsize_enwik9_m3_d64m.json
python3 tools/prof_collect_fullpmc.py gpurun_out/r04_m5_full_pmc profiles/r04_m5_fullsize_l2.md k_encode_runs_bt | tail -3
python - <<'PY'
import json
for t in ('r04_a','r04_m5','r04_m2','r04_p127'):
    d=json.load(open('profiles/%s_bench.json'%t)); r=d['roofline']; print(t, d['value'], r['avg_launch_ms'], r['frac'], (d.get('cpu_baseline') or {}).get('value'), d.get('p8_on_one_gpu',{}).get('value') if t=='r04_a' else '')
for f in ('r04_tree_bench','r04_tree_bench_cold'):
    d=json.load(open('profiles/%s.json'%f)); print(f, d['value'], d['bit_exact_vs_reference'], (d.get('cpu_baseline') or {}).get('value'), d['last_step_stats_rank0']['seconds_total'], d['last_step_stats_rank0']['seconds_encode'], d['last_step_stats_rank0']['seconds_setup'])
for f in ('r04_fullsize_enwik9_m3_d64m','r04_fullsize_silesia_m5_d256m','r04_fullsize_mix5_m2_d1024m_task0'):
    d=json.load(open('profiles/%s.json'%f)); print(f, d['encode_seconds'], d['MBps'], d['sha256']==d['reference']['sha256'])
PY
grep -o '"library_sha256_16": "[0-9a-f]*"' profiles/pmc_traffic.json | sort | uniq -c
sha256sum csc_amd/libcsc_mi355x.so | cut -c1-16
