#!/bin/bash
# Runs ON THE GPU BOX: the config-5-shape alignment kernel, both XCD placements, both flavours, several times (bench.py's own case loop).
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
sc5 = _hip.align_scratch(d5, n5, dev)
def t(fn, reps=15):
    for _ in range(40): fn()
    torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in e])) * 1e3
s = _hip.stream()
for rnd in range(3):
    row = {}
    rows_out = torch.empty(n5 * layer5.d_r, device=dev)
    os.environ["CVF_K1_XCD"] = "0"
    for tag, nopipe in (("pipe", None), ("slice", "1")):
        if nopipe is None:
            os.environ.pop("CVF_K1_NOPIPE", None)
            os.environ["CVF_K1_PIPE_MIN_GROUPS"] = "1024"   # (forced: by default it takes the flavours / sizes where it pays)
        else:
            os.environ.pop("CVF_K1_PIPE_MIN_GROUPS", None)
            os.environ["CVF_K1_NOPIPE"] = nopipe
        row["feat_" + tag] = round(t(lambda: lib.cvf_align_feature_fwd(d5, P(x5), n5, P(f_tmp), None, None, None, s)), 1)
        row["gen_" + tag] = round(t(lambda: lib.cvf_align_feature_fwd(d5, P(x5), n5, P(f_tmp), None, P(a_tmp), P(sc5), s)), 1)
        row["rows_" + tag] = round(t(lambda: lib.cvf_align_feature_fwd(d5, P(x5), n5, None, P(rows_out), None, None, s)), 1)
        row["c5step16k_" + tag] = round(t(lambda: lib.cvf_align_feature_fwd(d5, P(x5), 16000, P(f_tmp), None, P(a_tmp), P(sc5), s)), 1)
    os.environ.pop("CVF_K1_NOPIPE", None)
    os.environ["CVF_K1_PIPE_MIN_GROUPS"] = "1024"
    for pb in (1, 8):
        os.environ["CVF_K1_PIPE_PROBE"] = str(pb)
        row["feat_pipe_probe%d" % pb] = round(t(lambda: lib.cvf_align_feature_fwd(d5, P(x5), n5, P(f_tmp), None, None, None, s)), 1)
    for pb in (8,):
        os.environ["CVF_K1_PIPE_PROBE"] = str(pb)
        row["gen_pipe_probe%d" % pb] = round(t(lambda: lib.cvf_align_feature_fwd(d5, P(x5), n5, P(f_tmp), None, P(a_tmp), P(sc5), s)), 1)
    os.environ.pop("CVF_K1_PIPE_PROBE", None)
    os.environ["CVF_K1_XCD"] = "2"; os.environ["CVF_K1_NOPIPE"] = "1"
    row["feat_slice_streamonly"] = round(t(lambda: lib.cvf_align_feature_fwd(d5, P(x5), n5, P(f_tmp), None, None, None, s)), 1)
    os.environ["CVF_K1_XCD"] = "0"; os.environ.pop("CVF_K1_NOPIPE", None)
    n4 = x5.numel() // 4
    dst_r = torch.empty(int(lib.cvf_probe_stream_out_floats(1, n4)), device=dev)
    row["read_sweep_us"] = round(t(lambda: lib.cvf_probe_stream(1, P(dst_r), P(x5.reshape(-1)), n4, s)), 1)
    row["read_sweep_TBps"] = round(16.0 * n4 / row["read_sweep_us"] / 1e6, 2)
    print(json.dumps(row))
PY
