# round 5, GPU session 2: level-5 role functions + coder wavefront
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_s2
O=gpurun_out/r05_s2
timeout 900 python -m pytest tests/test_gpu_pipes.py -x -q -m gpu > $O/gputest_a.log 2>&1; tail -3 $O/gputest_a.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "5 or level or batch or smoke or golden" > $O/gputest_b.log 2>&1; tail -3 $O/gputest_b.log
AB_REPS=2 timeout 600 python tools/gpu_ab2.py m5 h1 h2 cur > $O/ab.txt 2>&1; cat $O/ab.txt
KIND=silesia DICT_MIB=256 timeout 300 python tools/gpu_timers.py 5 4 > $O/timers_m5.txt 2>&1; cat $O/timers_m5.txt
KIND=text DICT_MIB=256 timeout 300 python tools/gpu_timers.py 5 4 > $O/timers_m5_text.txt 2>&1; cat $O/timers_m5_text.txt
