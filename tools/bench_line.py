"""Print the interesting numbers of one bench.py JSON line: python tools/bench_line.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"], 4), "value %.4g" % d["value"], {k: round(v["ms"], 4) for k, v in d.get("per_stage_roofline", {}).items()})
dev = d.get("developed")
if dev:
    print("developed", round(dev["ms_per_step"], 4), {k: round(v, 4) for k, v in dev["stage_ms"].items()}, "overflow", dev.get("hit_list_overflow_fraction"))
