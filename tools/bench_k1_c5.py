import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "colvars-finder_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from bench_k1 import run, big_features
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
feats = big_features(5000, np.random.RandomState(3)) if mode == "full" else [("position", (17,))]
run(f"config5-shape 20k frames [{mode}]", 5000, 20_000, feats, reps=10)
