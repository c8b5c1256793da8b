for i in 1 2 3; do timeout 400 python bench.py --split 8 --steps 4 --warmup 1 --no-cpu-baseline --multi-streams "" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['bit_exact_vs_reference'], [t['task'] for t in d['tasks'] if not t['bit_exact_vs_reference']])
"; done
