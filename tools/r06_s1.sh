# round 6, GPU session 1: baseline of round 5's library on this box, node trace of the spine / edge wavefronts, section timers
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s1; mkdir -p $O
AB_REPS=2 timeout 600 python tools/gpu_ab2.py m3,p8 r5 > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt
timeout 300 python tools/gpu_dp4_trace.py 3000 3001 9000 > $O/trace.txt 2>&1; grep -v amdgpu.ids $O/trace.txt | head -70
timeout 300 python tools/gpu_dp4_timers.py 4 > $O/timers_m3_text.txt 2>&1; grep -v amdgpu.ids $O/timers_m3_text.txt | tail -22
