for t in 64 32 16 8; do
  CARETTA_TRACE_THREADS=$t python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > /tmp/b_$t.json
  python - $t <<'PY'
import sys, json
t = sys.argv[1]
d = json.loads(open(f"/tmp/b_{t}.json").read())
print("threads", t, round(d["value"]), round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["stage_ms"].items()})
PY
done
