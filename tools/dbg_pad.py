import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "colvars-finder_amd"))
from tests.synth import Traj, diag_coeff_for, make_molecule_traj
from colvarsfinder import core, nn, pp
from oracle import losses, nnref
from oracle.pp import AlignFeature
dev = torch.device("cuda:0")
for dims in ([30, 20, 10, 1], [30, 20, 20, 1], [30, 12, 12, 1], [30, 20, 20, 20, 1]):
    n_atoms, B, k = 10, 300, 2
    traj, w, ref = make_molecule_traj(n_atoms, B, seed=321)
    sd0 = nnref.init_eigenfunctions(dims, k, torch.Generator().manual_seed(9))
    model = nn.EigenFunctions(dims, k); model.load_state_dict(sd0)
    a = torch.tensor(diag_coeff_for(n_atoms, 2), dtype=torch.float32)
    layer = pp.AlignFeatureLayer(n_atoms, list(range(n_atoms)), ref, [("position", tuple(range(n_atoms)))]).to(dev)
    task = core.EigenFunctionTask(Traj(traj, w, 1.0), layer, model, "/tmp/dbg", 12.0, [1.0, 0.5], diag_coeff=a, beta=1.0, lag_tau=0,
                                  learning_rate=2e-3, k=k, batch_size=100, num_epochs=3, device=dev, verbose=False, save_model_every_step=0)
    task.loss_func(torch.tensor(traj), torch.tensor(w), None, None); task.backward()
    torch.set_default_dtype(torch.float64)
    sd = {n: p.double().requires_grad_(True) for n, p in sd0.items()}
    X = torch.tensor(traj, dtype=torch.float64, requires_grad=True)
    lo = losses.ef_loss(sd, k, AlignFeature(list(range(n_atoms)), ref, [("position", tuple(range(n_atoms)))], False), X, torch.tensor(w),
                        alpha=12.0, eig_w=[1.0, 0.5], diag_coeff=a.double(), beta=1.0)[0]
    lo.backward()
    torch.set_default_dtype(torch.float32)
    print(dims)
    for (n, p) in model.named_parameters():
        g, wv = p.grad.cpu().numpy(), sd[n].grad.numpy()
        print(f"  {n:28s} max|got-want| {np.abs(g - wv).max():10.3e}   max|want| {np.abs(wv).max():10.3e}")
