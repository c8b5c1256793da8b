# round 5, GPU session 5: where the pipeline form stops paying as streams outnumber CUs; decoder / hp timers of the current kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_s5
O=gpurun_out/r05_s5
MS_BYTES=1000000000 timeout 1500 python tools/gpu_many_streams.py 512 640 768 > $O/many_streams.txt 2>&1; grep -v amdgpu.ids $O/many_streams.txt
timeout 600 python tools/gpu_dec_timers.py > $O/dec_timers.txt 2>&1; grep -v amdgpu.ids $O/dec_timers.txt | head -40
