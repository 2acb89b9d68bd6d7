# developer: print the interesting fields of a bench.py JSON line (file given on the command line)
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(j["value"]), "ms_per_step", j["ms_per_step"], "repeat", j.get("repeat_ms_per_step"), "frac", round(j["roofline"]["frac"], 4),
      "floor", j["roofline"].get("latency_floor_us"))
cb = j.get("cpu_baseline") or {}
print("cpu", cb.get("value"), cb.get("leg"), cb.get("build"))
a = j.get("also") or {}
print({k: round(v.get("value") or v.get("us_per_call") or v.get("us_per_step") or 0, 1) for k, v in a.items() if isinstance(v, dict)})
print({k: v.get("parity_vs_fp64_oracle") for k, v in a.items() if isinstance(v, dict) and "parity_vs_fp64_oracle" in v})
for k in ("parity_vs_fp64_oracle", "parity", "dist"):
    if k in j: print(k, j[k])
