#!/usr/bin/env python3
"""Runs ON THE GPU BOX: the pipelined large-batch alignment kernel against the one-group-per-workgroup kernel, bit for bit, over random
molecule sizes / batch sizes / feature lists / output flavours, every case launched repeatedly (the roles of the pipelined kernel meet at
counters in LDS: a protocol error would show as a rare mismatch).   python tools/k1_pipe_fuzz.py [cases] [repeats]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "colvars-finder_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from colvarsfinder import _hip, pp  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 25
dev = torch.device("cuda:0")
lib, P, st = _hip.lib(), _hip.ptr, _hip.stream()
rs = np.random.RandomState(20261005)
bad = 0
for case in range(cases):
    n_atoms = int(rs.choice([400, 1000, 1600, 2600, 3500, 5000, 6000]))
    B = int(rs.choice([8192, 8200, 9001, 12345, 16384, 20000, 33333])) + int(rs.randint(0, 64))
    n_align = n_atoms if rs.rand() < 0.7 else 4 * int(rs.randint(n_atoms // 8, n_atoms // 4))
    angles = bool(rs.rand() < 0.3)
    n_pos, n_dih, n_bond, n_ang = int(rs.randint(8, 60)), int(rs.randint(60, 120)), int(rs.randint(10, 100)), int(rs.randint(0, 30))
    feats = [("position", tuple(int(i) for i in rs.choice(n_atoms, n_pos, replace=False)))]
    feats += [("dihedral", tuple(int(i) for i in rs.choice(n_atoms, 4, replace=False))) for _ in range(n_dih)]
    feats += [("bond", tuple(int(i) for i in rs.choice(n_atoms, 2, replace=False))) for _ in range(n_bond)]
    feats += [("angle", tuple(int(i) for i in rs.choice(n_atoms, 3, replace=False))) for _ in range(n_ang)]
    ref = rs.normal(scale=8.0, size=(n_atoms, 3))
    layer = pp.AlignFeatureLayer(n_atoms, list(range(n_align)), ref[:n_align], feats, angles).to(dev)
    d_r = layer.d_r
    if d_r < 272 or d_r > 384:
        continue
    g = torch.Generator(device=dev).manual_seed(case)
    x = (torch.tensor(ref, device=dev, dtype=torch.float32)[None] + 0.4 * torch.randn(B, n_atoms, 3, device=dev, generator=g)).contiguous()
    desc, T = layer.pp_desc(), _hip.ntiles(B)
    scratch = _hip.align_scratch(desc, B, dev)

    def run(flavour):
        tiled = torch.zeros(T * d_r * 64, device=dev)
        rows = torch.zeros(B * d_r, device=dev)
        aux = torch.zeros(T * 18 * 64, device=dev)
        if scratch is not None:
            scratch.zero_()
        args = dict(features=(P(tiled), None, None, None), generator=(P(tiled), None, P(aux), P(scratch)),
                    rows=(None, P(rows), None, None), both=(P(tiled), P(rows), None, None))[flavour]
        _hip.check(lib.cvf_align_feature_fwd(desc, P(x), B, *args, st), "k1")
        torch.cuda.synchronize()
        return [u.clone() for u in dict(features=[tiled], generator=[tiled, aux, scratch], rows=[rows], both=[tiled, rows])[flavour]]

    for flavour in ("features", "generator", "rows", "both"):
        os.environ["CVF_K1_NOPIPE"] = "1"
        want = run(flavour)
        del os.environ["CVF_K1_NOPIPE"]
        os.environ["CVF_K1_PIPE_MIN_GROUPS"] = "1024"
        miss = 0
        for _ in range(repeats):
            got = run(flavour)
            miss += int(not all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(got, want)))
        del os.environ["CVF_K1_PIPE_MIN_GROUPS"]
        bad += miss
        print(f"case {case}: atoms {n_atoms} align {n_align} frames {B} d_r {d_r} angles {angles} {flavour}: {miss} of {repeats} launches differ", flush=True)
print("TOTAL mismatching launches:", bad)
sys.exit(1 if bad else 0)
