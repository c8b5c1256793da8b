# round 5, GPU session 3: roles as functions of their own in the level-3 / level-1,2 forms too
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_s3
O=gpurun_out/r05_s3
timeout 1200 python -m pytest tests/test_gpu_pipes.py tests/test_gpu_parity.py -x -q -m gpu > $O/gputest_a.log 2>&1; tail -3 $O/gputest_a.log
AB_REPS=2 timeout 900 python tools/gpu_ab2.py m3,p8,m2,m5 r4 h3 cur > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 300 python tools/gpu_dp4_timers.py > $O/timers_m3.txt 2>&1; tail -40 $O/timers_m3.txt
KIND=mix5 DICT_MIB=1024 timeout 300 python tools/gpu_timers.py 2 4 > $O/timers_m2.txt 2>&1; cat $O/timers_m2.txt
