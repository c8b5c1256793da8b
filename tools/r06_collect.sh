# after `gpurun -- 'bash tools/r06_campaign_a.sh; bash tools/prof_dec.sh r06_dec'` and `gpurun -- bash tools/r06_campaign_b.sh`:
# everything under gpurun_out/ that is judged -> profiles/ (round 6, final library)
set -e
cd "$(dirname "$0")/.."
python tools/prof_collect.py r06_a | tail -4
python tools/prof_collect_cfg.py r06_m5 m5_d256m_single_stream "silesia stand-in -m5 -d256m, one stream, 2 chunks of 2 MiB" | tail -3
python tools/prof_collect_cfg.py r06_m2 m2_d1024m_single_stream "mix5 -m2 -d1024m, one stream, 2 chunks of 2 MiB" | tail -3
python tools/prof_collect_cfg.py r06_p127 m3_d64m_p127 "enwik9 stand-in -m3 -d64m as -p127 tasks, 2 chunk rounds of 127 x 2 MiB" | tail -3
python tools/prof_collect_cfg.py r06_p8 m3_d64m_p8 "enwik9 stand-in -m3 -d64m as -p8 tasks, 2 chunk rounds of 8 x 2 MiB" | tail -3
python tools/prof_collect_cfg.py r06_p954 m3_d64m_p954 "enwik9 stand-in -m3 -d64m as 954 task streams, one launch" | tail -3 || true
python tools/prof_collect_dec.py r06_dec | tail -3 || true
[ -f gpurun_out/r06_tree/bench.json ] && { python tools/prof_collect_tree.py r06_tree | tail -3 || true; }
for f in r06_timers_m3_text r06_timers_m5_silesia r06_timers_m2_mix5 r06_timers_m2_text r06_dec_timers r06_batch_probe r06_hp_text; do [ -f gpurun_out/$f.txt ] && cp gpurun_out/$f.txt profiles/$f.txt; done
[ -f gpurun_out/r06_dec_speed.txt ] && grep -v amdgpu.ids gpurun_out/r06_dec_speed.txt > profiles/r06_dec_speed.txt
[ -f gpurun_out/r06_full/fullsize.json ] && cp gpurun_out/r06_full/fullsize.json profiles/r06_fullsize_enwik9_m3_d64m.json
[ -f gpurun_out/r06_full/fullsize_silesia_m5_d256m.json ] && cp gpurun_out/r06_full/fullsize_silesia_m5_d256m.json profiles/r06_fullsize_silesia_m5_d256m.json
[ -f gpurun_out/r06_full/fullsize_mix5_m2_d1024m_task0.json ] && cp gpurun_out/r06_full/fullsize_mix5_m2_d1024m_task0.json profiles/r06_fullsize_mix5_m2_d1024m_task0.json
grep -o '"library_sha256_16": "[0-9a-f]*"' profiles/pmc_traffic.json | sort | uniq -c
sha256sum csc_amd/libcsc_mi355x.so | cut -c1-16
