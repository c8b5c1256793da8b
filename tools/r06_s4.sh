# round 6, GPU session 4: deferred match trees -- smoke, level-3 parity tests, A/B incl. loop-alignment variants
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s4; mkdir -p $O
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -4 $O/smoke.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream_matches or fresh_inputs or level3 or config2 or filters_off or wavefront" > $O/parity.txt 2>&1; tail -15 $O/parity.txt
AB_REPS=2 timeout 600 python tools/gpu_ab2.py m3,p8 r5 cur > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt
AB_REPS=2 timeout 600 python tools/gpu_ab2.py m3 al64 al128 al256 > $O/ab2.txt 2>&1; grep -v amdgpu.ids $O/ab2.txt
