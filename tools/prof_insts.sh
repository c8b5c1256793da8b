# instruction mix of the encode kernel (separate --pmc passes, program directly after `--`); run on the GPU box:
#   gpurun -- bash tools/prof_insts.sh <outdir> [MiB]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/insts}
MIB=${2:-4}
mkdir -p $OUT
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS"; do
  TAG=$(echo $SET | tr ' ' '_')
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc_$TAG -o p -- python3 tools/gpu_one.py 3 $MIB > $OUT/$TAG.log 2> $OUT/$TAG.err
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "encode_runs" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(f"{k:28s} {tot[k]:16.0f}")
PY
tail -2 $OUT/SQ_INSTS_VALU_SQ_INSTS_SALU.log
