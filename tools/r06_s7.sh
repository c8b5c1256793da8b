# round 6, GPU session 7: the one-wavefront form with the next node's gather asked for a node early -- parity of the forms that use it, the 954-stream line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream_matches or fresh_inputs or wavefront or fallbacks or custom_props or config3 or batch" > $O/parity.txt 2>&1; tail -4 $O/parity.txt
timeout 600 python bench.py --split 954 --steps 1 --warmup 0 --no-cpu-baseline > $O/p954.json 2> $O/p954.err; python - <<'PY'
import json
d=json.loads([x for x in open("gpurun_out/r06_s7/p954.json") if x.startswith("{")][-1]); print("p954", d["value"], d.get("bit_exact_vs_reference"), d["roofline"].get("kernel"))
PY
