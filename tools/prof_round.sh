# One round's measurements on the GPU box (gpurun -- bash tools/prof_round.sh r02_a):
#   bench line, rocprofv3 kernel stats of the same command, HBM traffic + instruction mix counters (separate --pmc passes,
#   --kernel-trace only, the program directly after `--`).  Everything lands in gpurun_out/<tag>/; tools/prof_summary.py
#   turns it into the files committed under profiles/.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py --steps 12 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 bench.py --steps 4 --warmup 1 --cpu-sample-mib 8 --multi-streams "" --p8-steps 0 --steady-steps 0 --other-steps 0 > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  T=$(echo $SET | tr ' ' '_')
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc_$T -o p -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --multi-streams "" --p8-steps 0 --steady-steps 0 --other-steps 0 > $OUT/pmc_$T.json 2> $OUT/pmc_$T.err
done
find $OUT -name "*.csv" | head -40
tail -c 400 $OUT/bench.json
