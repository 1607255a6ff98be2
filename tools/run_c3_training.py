#!/usr/bin/env python3
"""End-to-end run of the product path at BASELINE config 3: EigenFunctionTask.train() (split, static batches, train + test
loops, logging, save_model incl. the TorchScript export) on the synthetic 22-atom, 100 000-frame trajectory of bench.py.
Prints one JSON line: per-epoch mean losses (first / last), epoch wall time and the epoch-level frames/s (secondary
metric: comparable to the reference notebooks' tqdm rates)."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "colvars-finder_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from colvarsfinder import core, nn, pp  # noqa: E402
from tests.synth import Traj, diag_coeff_for  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
x, w, ref = bench.make_shard(100_000, 0)
a = torch.tensor(diag_coeff_for(bench.N_ATOMS, bench.SEED), dtype=torch.float32)
torch.manual_seed(bench.SEED)
np.random.seed(bench.SEED)
model = nn.EigenFunctions(bench.LAYERS, bench.K_NETS)
layer = pp.AlignFeatureLayer(bench.N_ATOMS, list(range(bench.N_ATOMS)), ref, [("position", tuple(range(bench.N_ATOMS)))])
with tempfile.TemporaryDirectory() as tmp:
    task = core.EigenFunctionTask(Traj(x, w, 1.0), layer, model, tmp, bench.ALPHA, bench.EIG_W, diag_coeff=a, beta=bench.BETA, lag_tau=0,
                                  learning_rate=bench.LR, k=bench.K_NETS, batch_size=20000, num_epochs=epochs, test_ratio=0.2,
                                  device=dev, verbose=False, save_model_every_step=epochs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    task.train()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    files = sorted(os.listdir(os.path.join(tmp, "latest")))
tr = np.stack([np.asarray(e[0]).mean(0) for e in task.loss_list])
te = np.stack([np.asarray(e[1]).mean(0) for e in task.loss_list])
assert np.isfinite(tr).all() and np.isfinite(te).all()
print(json.dumps(dict(run="EigenFunctionTask.train(), config 3 (22 atoms, 100k frames, k=3, B=20000, 80/20 split)", epochs=epochs,
                      steps_per_epoch=[len(task.loss_list[0][0]), len(task.loss_list[0][1])],
                      train_loss_first_last=[float(tr[0, 0]), float(tr[-1, 0])], test_loss_first_last=[float(te[0, 0]), float(te[-1, 0])],
                      eig_last=[float(v) for v in tr[-1, 3:]], wall_s=wall, s_per_epoch=wall / epochs,
                      epoch_frames_per_s=100_000 * epochs / wall, saved=files)))
