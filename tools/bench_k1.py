#!/usr/bin/env python3
"""Roofline of the align+feature kernel K1 (cvf_align_feature_fwd) alone, at whole-shard launch sizes.

  config-3 shape: 22 atoms, position features (d_r = 66), 100k / 1M frames      -> 532 B/frame
  config-5 shape: 5000 atoms, 32 positions + 96 dihedrals + 96 distances (d_r = 384) -> 61 540 B/frame
Prints one JSON line per case: achieved GB/s = algorithmic bytes / mean HIP-event time of the launch.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "colvars-finder_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from colvarsfinder import _hip, pp  # noqa: E402


def big_features(n_atoms, rs):
    feats = [("position", tuple(int(i) for i in rs.choice(n_atoms, 32, replace=False)))]
    feats += [("dihedral", tuple(int(i) for i in rs.choice(n_atoms, 4, replace=False))) for _ in range(96)]
    feats += [("bond", tuple(int(i) for i in rs.choice(n_atoms, 2, replace=False))) for _ in range(96)]
    return feats


def run(name, n_atoms, n_frames, feats, reps=20, slot_copy=True, with_aux=True):
    dev = torch.device("cuda")
    rs = np.random.RandomState(7)
    ref = rs.normal(scale=20.0 if n_atoms > 100 else 2.0, size=(n_atoms, 3))
    layer = pp.AlignFeatureLayer(n_atoms, list(range(n_atoms)), ref, feats).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    x = (torch.tensor(ref, device=dev, dtype=torch.float32)[None] +
         0.5 * torch.randn(n_frames, n_atoms, 3, device=dev, generator=g)).contiguous()
    T = _hip.ntiles(n_frames)
    feat = torch.empty(T * layer.d_r * 64, device=dev)
    aux = torch.empty(T * 18 * 64, device=dev)
    desc, lib = layer.pp_desc(), _hip.lib()
    scratch = _hip.align_scratch(desc, n_frames, dev) if slot_copy else None

    def launch():
        _hip.check(lib.cvf_align_feature_fwd(desc, _hip.ptr(x), n_frames, _hip.ptr(feat), None, _hip.ptr(aux) if with_aux else None, _hip.ptr(scratch),
                                             _hip.stream()), "k1")

    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        launch()
        b.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    bpf = 12 * n_atoms + 4 + 4 * layer.d_r
    gbs = bpf * n_frames / (ms * 1e-3) / 1e9
    print(json.dumps(dict(case=name, n_atoms=n_atoms, frames=n_frames, d_r=layer.d_r, bytes_per_frame=bpf, avg_launch_us=ms * 1e3,
                          achieved_GBps=gbs, frac_of_8TBps=gbs / 8000.0, frames_per_s=n_frames / (ms * 1e-3))))


if __name__ == "__main__":
    rs = np.random.RandomState(3)
    if "--c5" not in sys.argv:
      run("config3-shape 100k frames", 22, 100_000, [("position", tuple(range(22)))])
      run("config3-shape 1M frames", 22, 1_000_000, [("position", tuple(range(22)))])
      run("config3-shape 1M frames, features only (no rotation/centroid output)", 22, 1_000_000, [("position", tuple(range(22)))],
          with_aux=False)
    run("config5-shape 20k frames", 5000, 20_000, big_features(5000, rs))
    run("config5-shape 100k frames", 5000, 100_000, big_features(5000, rs), reps=5)
    # the same without the compact copy of the feature atoms (an extra OUTPUT of 12 n_slot B/frame that only the
    # generator-mode derivative kernel consumes; AutoEncoderTask / transfer mode do not ask for it)
    run("config5-shape 100k frames, features only (no slot copy)", 5000, 100_000, big_features(5000, rs), reps=5, slot_copy=False,
        with_aux=False)
