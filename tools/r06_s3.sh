# round 6, GPU session 3: node trace + section timers of the new straight-line paths (dev build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s3; mkdir -p $O
timeout 300 python tools/gpu_dp4_trace.py 3000 9000 > $O/trace.txt 2>&1; grep -v amdgpu.ids $O/trace.txt | head -90
timeout 300 python tools/gpu_dp4_timers.py 4 > $O/timers_m3_text.txt 2>&1; grep -v amdgpu.ids $O/timers_m3_text.txt | tail -16
