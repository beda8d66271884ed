"""prints the headline figures of a mirror_bench.py / bench.py output file (helper of the A/B command lines in gpurun calls)"""
import json, sys
txt = open(sys.argv[1]).read().strip()
try:
    d = json.loads(txt)
except ValueError:
    d = json.loads(txt.split("\n")[-1])
if "ms_per_step" in d:
    print("bench us/frame", round(d["ms_per_step"] * 1000, 1), "roofline", round(d["roofline"]["frac"], 3))
else:
    print([(k, v["deferred"]["us_per_keyframe"], v["deferred"]["host_us_in_calls"]["UpdateView"], v["synchronous"]["us_per_keyframe"])
           for k, v in d.items() if isinstance(v, dict) and isinstance(v.get("deferred"), dict)])
