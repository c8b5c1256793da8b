# round 5, GPU session 4: decoder scalars in SGPRs, level-5 ring of 256, many-streams probe, full GPU tier
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_s4
O=gpurun_out/r05_s4
AB_REPS=2 timeout 900 python tools/gpu_ab2.py m5,m3,m2,dec n3 b1 cur > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; tail -3 $O/gputest.log
KIND=silesia DICT_MIB=256 timeout 300 python tools/gpu_timers.py 5 4 > $O/timers_m5.txt 2>&1; grep -v amdgpu.ids $O/timers_m5.txt
timeout 600 python tools/gpu_dec_timers.py > $O/dec_timers.txt 2>&1; grep -v amdgpu.ids $O/dec_timers.txt | head -40
MS_BYTES=1000000000 timeout 900 python tools/gpu_many_streams.py 954 400 > $O/many_streams.txt 2>&1; grep -v amdgpu.ids $O/many_streams.txt
