"""How the timed region of bench.py behaves at the driver's K = 20 steps: repeated 20-step regions (4 epoch-graph
replays) after different pre-conditions, with host-side times of each replay call.  Run on the GPU box."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sys.path.insert(0, os.path.join(bench.ROOT if hasattr(bench, "ROOT") else os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "colvars-finder_amd"))
from colvarsfinder import core, nn, pp
from tests.synth import Traj, diag_coeff_for

dev = torch.device("cuda:0")
ref = np.random.RandomState(bench.SEED).normal(scale=2.0, size=(bench.N_ATOMS, 3))
a = torch.tensor(diag_coeff_for(bench.N_ATOMS, bench.SEED), dtype=torch.float32)
model = nn.EigenFunctions(bench.LAYERS, bench.K_NETS)
layer = pp.AlignFeatureLayer(bench.N_ATOMS, list(range(bench.N_ATOMS)), ref, [("position", tuple(range(bench.N_ATOMS)))])
tok = np.zeros((64, bench.N_ATOMS, 3), dtype=np.float32) + ref[None].astype(np.float32)
task = core.EigenFunctionTask(Traj(tok, np.ones(64), 1.0), layer, model, "/tmp/cvf_bench", bench.ALPHA, bench.EIG_W, diag_coeff=a,
                              beta=bench.BETA, lag_tau=0, learning_rate=bench.LR, k=bench.K_NETS, batch_size=20000, device=dev,
                              verbose=False, save_model_every_step=0)
X, Wt = bench.device_frames(100000, ref, 0.3, bench.SEED + 1, dev)
B, nb = 20000, 5
log = torch.zeros(nb, 3 + 2 * task.k, device=dev, dtype=torch.float64)
chunk = lambda: task._graph_call(("p", "chunk"), lambda: [task.train_step(X[b * B:(b + 1) * B], Wt[b * B:(b + 1) * B], out=log[b]) for b in range(nb)])
chunk(); chunk(); torch.cuda.synchronize()

def region(n_chunks=4):
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); ev0.record()
    hs = []
    for _ in range(n_chunks):
        h = time.perf_counter(); chunk(); hs.append((time.perf_counter() - h) * 1e6)
    ev1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return el * 1e6 / (n_chunks * nb), ev0.elapsed_time(ev1) * 1e3 / (n_chunks * nb), hs

for pre in (0, 0, 10, 50, 100, 300, 1000, 0, 0):
    for _ in range(pre):
        chunk()
    us, gpu_us, hs = region()
    print(f"pre-chunks {pre:5d}: wall {us:7.2f} us/step   events {gpu_us:7.2f} us/step   host per replay call {[round(h) for h in hs]}", flush=True)
    time.sleep(0.2)
print("-- back-to-back regions without sleep")
for _ in range(5):
    us, gpu_us, hs = region()
    print(f"wall {us:7.2f}  events {gpu_us:7.2f}  host {[round(h) for h in hs]}", flush=True)
print("-- 40 chunks per region")
for _ in range(3):
    us, gpu_us, hs = region(40)
    print(f"wall {us:7.2f}  events {gpu_us:7.2f}  host max {max(hs):.0f} mean {sum(hs)/len(hs):.0f}", flush=True)
print("-- hot GPU: 200 chunks, sync, one chunk, sync, then a 4-chunk region")
for _ in range(4):
    for _ in range(200):
        chunk()
    torch.cuda.synchronize(); chunk(); torch.cuda.synchronize()
    us, gpu_us, hs = region()
    print(f"wall {us:7.2f}  events {gpu_us:7.2f}  host {[round(h) for h in hs]}", flush=True)
print("-- hot GPU: 200 chunks, sync, three times (one chunk, sync), then a 4-chunk region")
for _ in range(4):
    for _ in range(200):
        chunk()
    for _ in range(3):
        torch.cuda.synchronize(); chunk()
    us, gpu_us, hs = region()
    print(f"wall {us:7.2f}  events {gpu_us:7.2f}  host {[round(h) for h in hs]}", flush=True)
