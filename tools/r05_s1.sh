# round 5, GPU session 1: tests on the new library, same-box A/B of the builds, the default bench line, level-5 section timers
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_s1
O=gpurun_out/r05_s1
timeout 900 python -m pytest tests/test_gpu_pipes.py tests/test_gpu_parity.py -x -q -m gpu > $O/gputest_a.log 2>&1; tail -3 $O/gputest_a.log
AB_REPS=2 timeout 600 python tools/gpu_ab2.py m3,p8,m5,m2 r4 u1 cur > $O/ab.txt 2>&1; cat $O/ab.txt
KIND=silesia DICT_MIB=256 timeout 300 python tools/gpu_timers.py 5 4 > $O/timers_m5.txt 2>&1; cat $O/timers_m5.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_pipes.py --deselect tests/test_gpu_parity.py > $O/gputest_b.log 2>&1; tail -3 $O/gputest_b.log
