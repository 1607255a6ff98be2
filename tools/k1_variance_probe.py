import os, sys, json
import numpy as np
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "colvars-finder_amd")):
    sys.path.insert(0, p)
import torch
import bench
from colvarsfinder import _hip, pp
n = 100_000
dev = torch.device("cuda:0")
na = bench.C5["n_atoms"]
ref = np.random.RandomState(bench.SEED).normal(scale=2.0, size=(na, 3))
layer = pp.AlignFeatureLayer(na, list(range(na)), ref, bench.c5_features(na)).to(dev)
desc = layer.pp_desc()
x, _ = bench.device_frames(n, ref, 0.05, bench.SEED + 78, dev, chunk=5000)
lib, P = _hip.lib(), _hip.ptr
T = _hip.ntiles(n)
feat = torch.empty(T * layer.d_r * 64, device=dev)
bpf = 12 * na + 4 + 4 * layer.d_r
def launch():
    _hip.check(lib.cvf_align_feature_fwd(desc, P(x), n, P(feat), None, None, None, _hip.stream()), "k1")
def run(sleep, reps=12, skip=3):
    evs = []
    for _ in range(reps):
        if sleep: torch.cuda._sleep(sleep)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in evs[skip:]]
    t = float(np.mean(ts))
    return round(bpf * n / (t * 1e-3) / 8e12, 4), round(min(ts) * 1e3), round(max(ts) * 1e3)
print("sleep 200k     ", run(200_000))
print("sleep 200k     ", run(200_000))
print("no sleep       ", run(0))
print("no sleep x40   ", run(0, 40, 10))
print("sleep 200k     ", run(200_000))
print("sleep 2M       ", run(2_000_000))
for _ in range(300): launch()
torch.cuda.synchronize()
print("after 300 launches, sleep 200k", run(200_000))
print("no sleep x40   ", run(0, 40, 10))
