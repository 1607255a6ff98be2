#!/bin/bash
# Runs ON THE GPU BOX: pipelined against one-group-per-workgroup alignment kernel at several batch sizes (config-5 shape), features only and generator outputs.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"
python - <<'PY'
import os, sys, json
sys.argv = ["bench.py"]
import bench, torch, numpy as np
from colvarsfinder import _hip, pp
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
lib, P = _hip.lib(), _hip.ptr
n5, na5 = 100_000, 5000
ref5 = np.random.RandomState(bench.SEED).normal(scale=2.0, size=(na5, 3))
layer5 = pp.AlignFeatureLayer(na5, list(range(na5)), ref5, bench.c5_features(na5)).to(dev)
d5 = layer5.pp_desc()
x5, _ = bench.device_frames(n5, ref5, 0.05, bench.SEED + 78, dev, chunk=5000)
T = _hip.ntiles(n5)
f_tmp = torch.empty(T * layer5.d_r * 64, device=dev); a_tmp = torch.empty(T * 18 * 64, device=dev)
rows_out = torch.empty(n5 * layer5.d_r, device=dev)
sc5 = _hip.align_scratch(d5, n5, dev)
s = _hip.stream()
def t(fn, reps=20):
    for _ in range(40): fn()
    torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in e])) * 1e3
for rnd in range(2):
    row = {}
    for B in (16000, 24000, 32768, 50000, 100000):
        for tag, env in (("pipe", {"CVF_K1_PIPE_MIN_GROUPS": "1024"}), ("slice", {"CVF_K1_NOPIPE": "1"}), ("slice_nostore", {"CVF_K1_NOPIPE": "1", "CVF_K1_XCD": "4"})):
            for k_ in ("CVF_K1_NOPIPE", "CVF_K1_XCD", "CVF_K1_PIPE_MIN_GROUPS"): os.environ.pop(k_, None)
            os.environ.update(env)
            off = [0]
            def mk(args):
                def fn():
                    o = off[0]; off[0] = ((o + B) % max(n5 - B, 1)) // 64 * 64 if B < n5 else 0
                    lib.cvf_align_feature_fwd(d5, P(x5[o:]), B, *args, s)
                return fn
            row["feat%d_%s" % (B, tag)] = round(t(mk((P(f_tmp), None, None, None))), 1)
            if tag != "slice_nostore":
                row["gen%d_%s" % (B, tag)] = round(t(mk((P(f_tmp), None, P(a_tmp), P(sc5)))), 1)
                row["rows%d_%s" % (B, tag)] = round(t(mk((None, P(rows_out), None, None))), 1)
    for k_ in ("CVF_K1_NOPIPE", "CVF_K1_XCD", "CVF_K1_PIPE_MIN_GROUPS"): os.environ.pop(k_, None)
    print(json.dumps(row))
PY
