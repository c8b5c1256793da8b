# round 6, GPU session 6: the whole GPU tier + the default bench line (driver's command) on the library with the exports map
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s6; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; tail -5 $O/gputest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
l=[x for x in open("gpurun_out/r06_s6/bench.json") if x.startswith("{")]
d=json.loads(l[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("summary"))
PY
