cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01_d
python bench.py > gpurun_out/r01_d/bench.json 2> gpurun_out/r01_d/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01_d/prof -o bench -- python3 bench.py --steps 4 --warmup 1 --cpu-sample-mib 8 > gpurun_out/r01_d/bench_under_rocprof.json 2> gpurun_out/r01_d/prof.err
for SET in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/r01_d/pmc_$SET -o p -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --multi-streams "" > gpurun_out/r01_d/pmc_$SET.json 2> gpurun_out/r01_d/pmc_$SET.err
done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/r01_d/pmc_TCC -o p -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --multi-streams "" > gpurun_out/r01_d/pmc_TCC.json 2> gpurun_out/r01_d/pmc_TCC.err
find gpurun_out/r01_d -name "*.csv" | head -30
tail -c 600 gpurun_out/r01_d/bench.json
