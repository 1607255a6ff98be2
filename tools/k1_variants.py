#!/usr/bin/env python3
"""Developer tool: the dipeptide-shape align+feature kernel at 4 M frames under the dispatch switches of csrc/k1_align.hip
(default = persistent streaming waves; CVF_K1_NOSTREAM = one wave per tile, not persistent; CVF_K1_QUAD=1 = four lanes per frame)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, json
sys.path.insert(0, "%s/tools"); import bench_k1
bench_k1.run("dipeptide 4M, features only", 22, 4_000_000, [("position", tuple(range(22)))], reps=12, with_aux=False)
bench_k1.run("dipeptide 4M, + rotation rows", 22, 4_000_000, [("position", tuple(range(22)))], reps=12, with_aux=True)
''' % ROOT
for env in ({}, {"CVF_K1_NOSTREAM": "1"}, {"CVF_K1_QUAD": "1"}):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", CODE], env=e, capture_output=True, text=True, timeout=300)
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            d = json.loads(ln)
            print(env or "default", d["case"], round(d["avg_launch_us"], 1), "us", round(d["frac_of_8TBps"], 3))
    if r.returncode:
        print(env, "failed", r.stderr[-500:])
