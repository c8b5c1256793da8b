# round 6, GPU session 5: section timers of the coarse development build (per-window sections only: its speed is the product's to a per cent or two)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s5; mkdir -p $O
timeout 300 python tools/gpu_dp4_timers.py 4 > $O/timers_m3_text.txt 2>&1; grep -v amdgpu.ids $O/timers_m3_text.txt
