for h in 480 240 120 60; do
  python bench.py --height $h --steps 4 --warmup 1 --cpu-rows 0 2>/dev/null | tail -1 > /tmp/o.json
  python - $h <<'PY'
import sys, json
d = json.load(open('/tmp/o.json')); k = d["kernels_ms"]
print(sys.argv[1], round(d["ms_per_step"], 2), round(sum(k.values()), 2), {a: round(b, 2) for a, b in k.items()})
PY
done
