#!/bin/bash
# same-box alternating A/B of the bench line: tools/ab_bench.sh <out> <n rounds> "<env A>" "<env B>" [bench args...]
out=$1; n=$2; A=$3; B=$4; shift 4
mkdir -p $(dirname $out)
: > $out
for i in $(seq 1 $n); do
  for v in "$A" "$B"; do
    line=$(env $v python bench.py --no-cpu-baseline --no-pmc "$@" 2>/dev/null | tail -1)
    python - "$v" "$line" >> $out <<'PY'
import json, sys
v, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    legs = {k: round(x.get("value", 0), 2) for k, x in d.items() if isinstance(x, dict) and "value" in x and k not in ("roofline", "cpu_baseline")}
    print(f"{v:28s} value {d['value']:.2f} {d['unit']}  ms/step {d['ms_per_step']:.4f}  legs {legs}")
except Exception as e:
    print(v, "FAILED", e, line[:200])
PY
  done
done
cat $out
