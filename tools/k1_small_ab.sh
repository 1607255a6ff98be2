#!/bin/bash
# Runs ON THE GPU BOX: the config-5-shape alignment kernel at the step's batch sizes, pipelined kernel (forced with single groups) against the one-group-per-workgroup kernel.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"
python - <<'PY'
import os, sys, json
sys.argv = ["bench.py"]
import bench, torch, numpy as np
from colvarsfinder import _hip, pp
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
lib, P = _hip.lib(), _hip.ptr
n5, na5 = 32_000, 5000
ref5 = np.random.RandomState(bench.SEED).normal(scale=2.0, size=(na5, 3))
layer5 = pp.AlignFeatureLayer(na5, list(range(na5)), ref5, bench.c5_features(na5)).to(dev)
d5 = layer5.pp_desc()
x5, _ = bench.device_frames(n5, ref5, 0.05, bench.SEED + 78, dev, chunk=4000)
T = _hip.ntiles(n5)
f_tmp = torch.empty(T * layer5.d_r * 64, device=dev); a_tmp = torch.empty(T * 18 * 64, device=dev)
sc5 = _hip.align_scratch(d5, n5, dev)
s = _hip.stream()
def t(fn, reps=30):
    for _ in range(60): fn()
    torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in e])) * 1e3
for rnd in range(2):
    row = {}
    for B in (1000, 2000, 4000, 6000, 8000, 12000):
        # (rotate the start so that consecutive launches do not find their frames in the Infinity Cache)
        for tag, env in (("pipe", {"CVF_K1_PIPE_MIN_GROUPS": "0"}), ("slice", {"CVF_K1_NOPIPE": "1"})):
            for k_ in ("CVF_K1_PIPE_MIN_GROUPS", "CVF_K1_NOPIPE"): os.environ.pop(k_, None)
            os.environ.update(env)
            off = [0]
            def fn():
                o = off[0]; off[0] = (o + B) % (n5 - B + 1) // 64 * 64
                lib.cvf_align_feature_fwd(d5, P(x5[o:]), B, P(f_tmp), None, P(a_tmp), P(sc5), s)
            row["gen%d_%s" % (B, tag)] = round(t(fn), 1)
    print(json.dumps(row))
PY
