# Decoder profile (gpurun -- bash tools/prof_dec.sh <tag>): one 4 MiB level-3 text stream decoded by k_decode_run under rocprofv3
# --kernel-trace --stats, then separate --pmc passes (the program directly after `--`).  tools/prof_collect_dec.py -> profiles/<tag>_{kernel_stats,pmc}.md
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r05_dec}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 tools/gpu_dec_one.py 3 4 text > $OUT/plain.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o dec -- python3 tools/gpu_dec_one.py 3 4 text > $OUT/under_rocprof.txt 2> $OUT/prof.err
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  T=$(echo $SET | tr ' ' '_')
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc_$T -o p -- python3 tools/gpu_dec_one.py 3 4 text > $OUT/pmc_$T.txt 2> $OUT/pmc_$T.err
done
grep -v amdgpu.ids $OUT/plain.txt $OUT/under_rocprof.txt
