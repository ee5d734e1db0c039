"""Print the interesting numbers of one bench.py JSON line: python tools/bench_line.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"], 4), "value %.4g" % d["value"], {k: round(v["ms"], 4) for k, v in d.get("per_stage_roofline", {}).items()})
dev = d.get("developed")
if dev and "stage_ms" in dev:
    print("developed", round(dev["ms_per_step"], 4), {k: round(v, 4) for k, v in dev["stage_ms"].items()}, "overflow", (dev.get("at_end") or {}).get("hit_list_overflow_fraction"))
for k in ("resting",):
    r = d.get(k)
    if r:
        print(k, round(r["ms_per_step"], 4), {a: round(b, 4) for a, b in r["stage_ms"].items()})
print("cfl_ok", d.get("cfl_ok"), "spin_up", d["config"].get("spin_up_steps"), "dt", d["config"].get("dt"), "roofline frac", round(d["roofline"]["frac"], 4),
      "whole-step frac", round(d["roofline"]["whole_step"].get("frac", d["roofline"]["whole_step"].get("frac_per_gpu", 0.0)), 4))
