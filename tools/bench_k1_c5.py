#!/usr/bin/env python3
"""Align+feature kernel alone at BASELINE config 5's shape (5000 atoms, d_r = 384), 100 000 frames resident (6.2 GB per launch:
out of every cache): HIP-event time per launch, features only and with the generator-mode extras.  CVF_K1_NT=0 switches the
non-temporal coordinate loads off (developer comparison).   python tools/bench_k1_c5.py [frames]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "colvars-finder_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from colvarsfinder import _hip, pp  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dev = torch.device("cuda:0")
na = bench.C5["n_atoms"]
ref = np.random.RandomState(bench.SEED).normal(scale=2.0, size=(na, 3))
layer = pp.AlignFeatureLayer(na, list(range(na)), ref, bench.c5_features(na)).to(dev)
desc = layer.pp_desc()
x, _ = bench.device_frames(n, ref, 0.05, bench.SEED + 78, dev, chunk=5000)
lib, P = _hip.lib(), _hip.ptr
T = _hip.ntiles(n)
feat = torch.empty(T * layer.d_r * 64, device=dev)
aux = torch.empty(T * 18 * 64, device=dev)
scr = _hip.align_scratch(desc, n, dev)
bpf = 12 * na + 4 + 4 * layer.d_r


def run(with_extras, reps=12):
    evs = []
    for _ in range(reps):
        torch.cuda._sleep(200_000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _hip.check(lib.cvf_align_feature_fwd(desc, P(x), n, P(feat), None, P(aux) if with_extras else None,
                                             P(scr) if with_extras else None, _hip.stream()), "k1")
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    t = float(np.mean([a.elapsed_time(b) for a, b in evs[3:]]))
    return dict(avg_launch_us=t * 1e3, GBps=bpf * n / (t * 1e-3) / 1e9, frac_of_8TBps=bpf * n / (t * 1e-3) / 8e12)


print(json.dumps(dict(case="config-5 shape", frames=n, bytes_per_frame=bpf, nt=os.environ.get("CVF_K1_NT", "1"),
                      features_only=run(False), generator_outputs=run(True))))
