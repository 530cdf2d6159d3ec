#!/usr/bin/env python3
"""stdout of `oracle/_ref/ref_tool` running `rep N / op steps 1` from step 21 (the reference's own time loop, bench.py's workload)
-> profiles/r03/reference_window_<size>.json, the file bench.py's config.ref_iters_per_step quotes.  Works on a run that is still
going: whatever steps have finished are recorded.

    python scripts/reference_window_to_json.py <stdout.log> <size> <threads>
"""
import json
import os
import sys

log, size, threads = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps = []
for line in open(log):
    if line.startswith("REF steps"):
        kv = dict(p.split("=") for p in line.split()[2:])
        steps.append({"step": 21 + len(steps), "iters": int(float(kv["iters"])), "dt": float(kv["value"]), "seconds": float(kv["seconds"])})
out = {"size": size, "threads": threads, "first_step": 21, "workload": "taylor-green all-wall, nu=0.01, CFL=0.3, rampup=0 (bench.py's)",
       "steps": steps, "complete": len(steps) >= 25,
       "host": "build container (8 cores), not the GPU box: only the iteration counts are quoted, never the seconds"}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03", f"reference_window_{size}.json")
json.dump(out, open(path, "w"), indent=1)
print(path, len(steps), "steps", [s["iters"] for s in steps])
