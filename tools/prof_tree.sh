# The many-task line's HBM traffic (gpurun -- bash tools/prof_tree.sh <tag>): `bench.py --workload tree --steps 1 --warmup 0` under
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / L2 hits (separate passes, --kernel-trace only, the program directly after `--`).
# tools/prof_collect_tree.py turns gpurun_out/<tag>/ into profiles/<tag>_pmc.md and the `tree_m3_d64m` key of profiles/pmc_traffic.json.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r05_tree}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py --workload tree --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --workload tree --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_cold.json 2> $OUT/bench_cold.err
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $SET | tr ' ' '_')
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc_$T -o p -- python3 bench.py --workload tree --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$T.json 2> $OUT/pmc_$T.err
done
tail -c 400 $OUT/bench.json
