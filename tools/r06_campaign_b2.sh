# round-end measurements, part B, after the last fix of the level-3 form (gpurun -- bash tools/r06_campaign_b2.sh): the tree, level-3 timers, decoder speeds, text at
# levels 1 / 2, and the full-size pins of the two configurations whose code could have moved (enwik9 -m3: the fixed kernel; silesia -m5 once more).  The mix5 -m2 task-0
# pin (9.7 min) was run on library a8ef1d02d39f70ec; the level-1/2/5 sources did not change after it.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/prof_tree.sh r06_tree > gpurun_out/r06_tree.log 2>&1
timeout 300 python tools/gpu_dp4_timers.py 4 text > gpurun_out/r06_timers_m3_text.txt 2>&1
timeout 600 python tools/gpu_dec_speed.py > gpurun_out/r06_dec_speed.txt 2>&1
timeout 300 python tools/gpu_hp.py > gpurun_out/r06_hp_text.txt 2>&1
mkdir -p gpurun_out/r06_full
timeout 900 python tools/gpu_fullsize.py gpurun_out/r06_full > gpurun_out/r06_full_m3.log 2>&1
timeout 900 python tools/gpu_fullsize_cfg.py silesia_m5_d256m gpurun_out/r06_full > gpurun_out/r06_full_silesia.log 2>&1
python -c "
import json
d=json.load(open('gpurun_out/r06_tree/bench.json')); print('tree', d['value'], d['bit_exact_vs_reference'], d['cpu_baseline']['value'])
for f in ('fullsize','fullsize_silesia_m5_d256m'):
    d=json.load(open('gpurun_out/r06_full/%s.json'%f)); print(f, d['encode_seconds'], d['MBps'], d['sha256'][:16], d['reference']['sha256'][:16], d.get('bit_exact_vs_reference'))
"
