"""GPU (-m gpu): the reference's CALL CONTRACT on the MI355X path, and the rows of SURVEY section 8f that surround the step.

The reference's users evaluate the learned collective variables on CPU tensors made from the trajectory and take
``.detach().numpy()`` of the result (examples/2d/2d.ipynb:437-446, examples/dipeptide/main.ipynb:561-562); its training
loops hand ``self.colvar_model()`` to a user ``plot_class`` (core.py:530-532, 720-722); its tasks expose a
``torch.optim`` object as ``task.optimizer`` (core.py:163-166) and restart from ``load_model_filename``
(core.py:156-161).  These tests make exactly those calls (own code; nothing copied from the notebooks).
"""

import numpy as np
import pytest
import torch

from tests.synth import Traj, diag_coeff_for, make_2d_traj, make_molecule_traj

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _restore_dtype():
    yield
    torch.set_default_dtype(torch.float32)


def _position_layer(n_atoms, ref, dev):
    from colvarsfinder import pp
    return pp.AlignFeatureLayer(n_atoms, list(range(n_atoms)), ref, [("position", tuple(range(n_atoms)))]).to(dev)


class _Recorder:
    """A user plot class: evaluates the CV model the way the notebooks' plotting cells do - CPU tensor in, numpy out."""

    def __init__(self, traj, with_reg=False):
        self.X = torch.tensor(traj)            # CPU tensor, default dtype cast as torch.tensor does
        self.calls = []
        self.with_reg = with_reg

    def plot(self, cv_model, reg_model=None, epoch=0):
        out = cv_model(self.X).detach().numpy()
        assert isinstance(out, np.ndarray) and out.shape[0] == self.X.shape[0]
        if self.with_reg:
            assert reg_model is not None
            r = reg_model(self.X).detach().numpy()
            assert r.shape[0] == self.X.shape[0]
        self.calls.append((epoch, out))


def _ef_task(dev, tmp_path, n_atoms=10, n_frames=400, k=2, lag_tau=0.0, **kw):
    from colvarsfinder import core, nn
    traj, w, ref = make_molecule_traj(n_atoms, n_frames, seed=901)
    layer = _position_layer(n_atoms, ref, dev)
    torch.manual_seed(11)
    model = nn.EigenFunctions([3 * n_atoms, 20, 20, 20, 1], k)
    a = torch.tensor(diag_coeff_for(n_atoms, 4), dtype=torch.float32)
    args = dict(diag_coeff=a if lag_tau == 0 else None, beta=1.0, lag_tau=lag_tau, learning_rate=1e-3, k=k, batch_size=100,
                num_epochs=3, device=dev, verbose=False, save_model_every_step=0)
    args.update(kw)
    task = core.EigenFunctionTask(Traj(traj, w, 0.5), layer, model, str(tmp_path), 10.0, [1.0, 0.6][:k], **args)
    return task, traj, w


def test_colvar_model_takes_cpu_tensors_like_the_notebooks(dev, tmp_path):
    """EigenFunctionTask: ``cv = task.colvar_model(); cv(torch.tensor(traj)).detach().numpy()`` and the plot callback
    inside train() (core.py:530-532), generator and transfer mode."""
    for lag_tau in (0.0, 1.0):
        rec_traj = make_molecule_traj(10, 400, seed=901)[0]
        plotter = _Recorder(rec_traj)
        np.random.seed(3)
        task, traj, w = _ef_task(dev, tmp_path, lag_tau=lag_tau, plot_class=plotter, plot_frequency=1)
        task.train()
        assert [e for e, _ in plotter.calls] == [0, 1, 2]
        cv_model = task.colvar_model()
        x_cpu = torch.tensor(traj)                       # what the notebooks build from traj_obj.trajectory
        out = cv_model(x_cpu)
        assert out.device.type == "cpu" and out.dtype == x_cpu.dtype and tuple(out.shape) == (400, 2)
        got = out.detach().numpy()
        on_gpu = cv_model(x_cpu.to(dev))                 # device tensors keep working and stay on the device
        assert on_gpu.device.type == "cuda"
        np.testing.assert_allclose(got, on_gpu.detach().cpu().numpy(), rtol=0, atol=0)
        np.testing.assert_allclose(plotter.calls[-1][1], got, rtol=0, atol=0)   # the last callback saw the final model
        # float64 input (text trajectories, np.loadtxt): answered in float64
        assert cv_model(torch.tensor(traj, dtype=torch.float64)).dtype == torch.float64
        # the layer alone, as a user would call pp_layer(x)
        feats = task.preprocessing_layer(x_cpu)
        assert feats.device.type == "cpu" and tuple(feats.shape) == (400, 30)


def test_autoencoder_and_identity_layer_cpu_calls(dev, tmp_path):
    """AutoEncoderTask with the 2-D examples' Identity layer and with the alignment layer: colvar_model() on CPU tensors,
    plot callback inside train() (core.py:720-722)."""
    from colvarsfinder import core, nn
    x2d, w2d = make_2d_traj(600, seed=5, dtype=np.float64)
    plotter = _Recorder(x2d)
    torch.manual_seed(2)
    model = nn.AutoEncoder([2, 20, 20, 1], [1, 20, 20, 2])
    np.random.seed(8)
    task = core.AutoEncoderTask(Traj(x2d, w2d, 0.1), torch.nn.Identity(), model, str(tmp_path), learning_rate=5e-3, batch_size=200,
                                num_epochs=2, device=dev, verbose=False, save_model_every_step=0, plot_class=plotter,
                                plot_frequency=1)
    task.train()
    assert len(plotter.calls) == 2
    cv = task.colvar_model()(torch.tensor(x2d)).detach().numpy()      # float64 CPU in -> float64 numpy out
    assert cv.shape == (600, 1) and cv.dtype == np.float64
    np.testing.assert_allclose(cv, plotter.calls[-1][1], rtol=0, atol=0)

    traj, w, ref = make_molecule_traj(10, 300, seed=77)
    torch.manual_seed(2)
    model = nn.AutoEncoder([30, 20, 2], [2, 20, 30])
    task = core.AutoEncoderTask(Traj(traj, w, 1.0), _position_layer(10, ref, dev), model, str(tmp_path), batch_size=100, num_epochs=1,
                                device=dev, verbose=False, save_model_every_step=0)
    task.train()
    out = task.colvar_model()(torch.tensor(traj)).detach().numpy()
    assert out.shape == (300, 2) and np.isfinite(out).all()


def test_regautoencoder_models_take_cpu_tensors(dev, tmp_path):
    """RegAutoEncoderTask: plot_class.plot(colvar_model(), reg_model(), epoch=) (core.py:1136-1140) with CPU evaluation."""
    from colvarsfinder import core, nn
    traj, w, ref = make_molecule_traj(10, 400, seed=12)
    plotter = _Recorder(traj, with_reg=True)
    torch.manual_seed(4)
    model = nn.RegAutoEncoder([30, 20, 2], [2, 20, 30], [2, 20, 1], 2)
    np.random.seed(1)
    task = core.RegAutoEncoderTask(Traj(traj, w, 1.0), _position_layer(10, ref, dev), model, str(tmp_path), eig_weights=[1.0, 0.5],
                                   batch_size=100, num_epochs=2, alpha=1.0, gamma=[1.0, 5.0], lag_tau_ae=1.0, lag_tau_reg=1.0,
                                   device=dev, verbose=False, save_model_every_step=0, plot_class=plotter, plot_frequency=1)
    task.train()
    assert len(plotter.calls) == 2
    r = task.reg_model()(torch.tensor(traj)).detach().numpy()
    assert r.shape == (400, 2) and np.isfinite(r).all()


def test_optimizer_surface_lr_change_reaches_captured_graphs(dev, tmp_path):
    """``task.optimizer.param_groups[0]['lr']`` changed between epochs (a scheduler / manual decay) must take effect although
    every step after the first epoch is a hipGraph replay: the kernels read the rate from a device scalar."""
    snaps = {}

    class Decay:
        def plot(self, cv_model, epoch=0):
            snaps[epoch] = torch.cat([p.detach().reshape(-1) for p in task.model.parameters()]).cpu().clone()
            if epoch == 1:
                task.optimizer.param_groups[0]["lr"] = 0.0     # from here on the parameters must stand still

    np.random.seed(5)
    task, _, _ = _ef_task(dev, tmp_path, num_epochs=4, plot_class=Decay(), plot_frequency=1)
    assert task._use_graphs
    pg = task.optimizer.param_groups[0]
    assert pg["lr"] == 1e-3 and pg["betas"] == (0.9, 0.999) and pg["eps"] == 1e-8 and len(pg["params"]) == len(list(task.model.parameters()))
    task.train()
    assert not torch.equal(snaps[0], snaps[1])              # lr = 1e-3: moving
    assert torch.equal(snaps[1], snaps[2]) and torch.equal(snaps[2], snaps[3])   # lr = 0 inside replayed graphs


def test_optimizer_step_honours_edited_grads(dev, tmp_path):
    """loss_func -> backward() -> (user edits p.grad, e.g. clipping) -> optimizer.step(): the edit counts, as with torch.optim."""
    task, traj, w = _ef_task(dev, tmp_path)
    before = torch.cat([p.detach().reshape(-1) for p in task.model.parameters()]).clone()
    task.loss_func(torch.tensor(traj), torch.tensor(w), None, None)
    task.backward()
    assert all(p.grad is not None and float(p.grad.abs().sum()) >= 0 for p in task.model.parameters())
    for p in task.model.parameters():
        p.grad.zero_()
    task.optimizer.step()
    after = torch.cat([p.detach().reshape(-1) for p in task.model.parameters()])
    assert torch.equal(before, after)                       # zero gradient, zero moments: Adam leaves the parameters alone
    task.loss_func(torch.tensor(traj), torch.tensor(w), None, None)
    task.backward()
    task.optimizer.step()
    assert not torch.equal(before, torch.cat([p.detach().reshape(-1) for p in task.model.parameters()]))
    task.optimizer.zero_grad()
    assert all(p.grad is None for p in task.model.parameters())


def test_restart_from_model_file_and_optimizer_state(dev, tmp_path):
    """core.py:156-161 ``load_model_filename`` (+ the optimizer state the reference does not checkpoint):
    3 epochs, save, new task restarted from the files, 2 more epochs == 5 epochs in one go (same batches)."""
    import os

    def run(path, epochs, load=None, opt_state=None):
        np.random.seed(21)                                   # train() draws the split from NumPy's global RNG
        task, _, _ = _ef_task(dev, path, num_epochs=epochs, load_model_filename=load)
        if opt_state is not None:
            task.optimizer.load_state_dict(opt_state)
        task.train()
        return task

    full = run(tmp_path / "full", 5)
    part = run(tmp_path / "part", 3)
    part.save_model(2)
    fname = str(tmp_path / "part" / "latest" / "model.pt")
    assert os.path.isfile(fname)
    state = part.optimizer.state_dict()
    assert state["step"] == 3 * 3 and state["exp_avg"].device.type == "cpu"      # 320 train frames / 100 = 3 steps per epoch
    torch.save(state, str(tmp_path / "opt.pt"))
    rest = run(tmp_path / "rest", 2, load=fname, opt_state=torch.load(str(tmp_path / "opt.pt")))
    # the restarted model picks up the saved parameters ...
    got0 = torch.load(fname)
    assert set(got0) == set(full.model.state_dict())
    # ... and, with the moments and the step number restored, continues exactly where the first run stood
    for (n, a), (_, b) in zip(full.model.state_dict().items(), rest.model.state_dict().items()):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-6, atol=1e-7, err_msg=n)
    for e in range(2):
        np.testing.assert_allclose(rest.loss_list[e][0].numpy(), full.loss_list[3 + e][0].numpy(), rtol=1e-6)
    # without the optimizer state the restart is a different trajectory (Adam's bias correction starts over)
    cold = run(tmp_path / "cold", 2, load=fname)
    assert not np.allclose(cold.loss_list[1][0].numpy(), full.loss_list[4][0].numpy(), rtol=1e-6)
    # a missing file is reported, not fatal (core.py:160-161)
    run(tmp_path / "none", 1, load=str(tmp_path / "does_not_exist.pt"))


def test_unsupported_net_shape_is_rejected_at_construction(dev, tmp_path):
    from colvarsfinder import core, nn
    traj, w, ref = make_molecule_traj(10, 100, seed=1)
    # too wide; six hidden layers; ONE hidden layer wider than the one-hidden-layer kernels' widest (crashed in min())
    for dims in ([30, 80, 80, 1], [30, 20, 20, 20, 20, 20, 20, 1], [30, 24, 1], [30, 32, 1]):
        model = nn.EigenFunctions(dims, 2)
        with pytest.raises(NotImplementedError, match="no kernel instance"):
            core.EigenFunctionTask(Traj(traj, w, 1.0), _position_layer(10, ref, dev), model, str(tmp_path), 10.0, [1.0, 0.5], k=2,
                                   device=dev, verbose=False)
    with pytest.raises(NotImplementedError, match="Tanh"):
        core.EigenFunctionTask(Traj(traj, w, 1.0), _position_layer(10, ref, dev), nn.EigenFunctions([30, 20, 1], 1, torch.nn.GELU()),
                               str(tmp_path), 10.0, [1.0], k=1, device=dev, verbose=False)


@pytest.mark.parametrize("kind", ["f32", "f64", "strided", "small"])
def test_upload_matches_tensor_to(dev, kind):
    """the trajectory's way into HBM (page-locked single DMA for fp32, chunked convert-and-copy through two
    pinned staging buffers otherwise) is byte-equal to ``torch.as_tensor(a).to(float32).to(device)``."""
    from colvarsfinder import _hip
    rs = np.random.RandomState(0)
    if kind == "f32":
        a = rs.standard_normal((150_000, 22, 3)).astype(np.float32)            # 39.6 MB: the hipHostRegister path
    elif kind == "f64":
        a = rs.standard_normal((150_000, 22, 3))                                  # fp64 source: chunked conversion
    elif kind == "strided":
        a = rs.standard_normal((150_000, 44, 3)).astype(np.float32)[:, ::2, :]  # non-contiguous fp32
    else:
        a = rs.standard_normal((100, 22, 3))
    torch.cuda.synchronize()
    got = _hip.upload_f32(a, dev, chunk_bytes=8 << 20) if kind != "small" else _hip.upload_f32(a, dev)
    want = torch.as_tensor(np.ascontiguousarray(a)).to(torch.float32).to(dev)
    assert got.dtype == torch.float32 and got.is_contiguous() and tuple(got.shape) == tuple(a.shape)
    assert torch.equal(got, want)


class _AtomGroup:
    """Duck-typed MDAnalysis AtomGroup: the two attributes the constructors read."""

    def __init__(self, ix, positions=None):
        self.ix = np.asarray(ix)
        self.positions = positions

    def __len__(self):
        return len(self.ix)


def test_molann_style_layer_runs_on_the_gpu_vs_oracle(dev):
    """PreprocessingANN(AlignmentLayer(...), FeatureLayer(...)) built the way main.ipynb:333-348 builds it
    (global atom indices, positions of the align group), run on the GPU, against the CPU oracle on local indices."""
    from colvarsfinder import pp
    from oracle.pp import AlignFeature
    glob = [1, 4, 5, 6, 8, 10, 14, 15, 16, 18]
    traj, _, ref = make_molecule_traj(10, 500, seed=44)
    input_ag = _AtomGroup(glob, ref.astype(np.float32))
    align_ag = _AtomGroup(glob[:7], ref[:7].astype(np.float32))
    feats = [pp.Feature("p1", "position", input_ag), pp.Feature("d1", "dihedral", _AtomGroup([4, 6, 8, 14])),
             pp.Feature("a1", "angle", _AtomGroup([5, 6, 8])), pp.Feature("b1", "bond", _AtomGroup([1, 18]))]
    layer = pp.PreprocessingANN(pp.AlignmentLayer(align_ag, input_ag), pp.FeatureLayer(feats, input_ag)).to(dev)
    got = layer(torch.tensor(traj)).numpy()                  # CPU in, CPU out
    local = {g: i for i, g in enumerate(glob)}
    spec_feats = [("position", tuple(range(10))), ("dihedral", tuple(local[g] for g in [4, 6, 8, 14])),
                  ("angle", tuple(local[g] for g in [5, 6, 8])), ("bond", (local[1], local[18]))]
    torch.set_default_dtype(torch.float64)
    want = AlignFeature(list(range(7)), ref[:7].astype(np.float32).astype(np.float64), spec_feats, False)(
        torch.tensor(traj, dtype=torch.float64)).numpy()
    assert got.shape == want.shape == (500, 34)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6 * np.abs(want).max())


def test_nonuniform_hidden_widths_are_padded_and_match_the_oracle(dev, tmp_path):
    """The reference takes any layer_dims (nn.py:29-59); the kernels are instantiated for one hidden width per net.  Nets with
    other / mixed hidden widths (here [30, 20, 10, 1]) are laid out zero-padded to the next kernel width: loss, eigenvalues
    and every parameter gradient must equal the fp64 oracle's on the UNPADDED nets, the padding must stay exactly zero
    through training, and state_dict() must keep the shapes the user built."""
    from colvarsfinder import core, nn
    from oracle import losses, nnref
    from oracle.pp import AlignFeature
    n_atoms, B, k, dims = 10, 300, 2, [30, 20, 10, 1]
    traj, w, ref = make_molecule_traj(n_atoms, B, seed=321)
    sd0 = nnref.init_eigenfunctions(dims, k, torch.Generator().manual_seed(9))
    model = nn.EigenFunctions(dims, k)
    model.load_state_dict(sd0)
    a = torch.tensor(diag_coeff_for(n_atoms, 2), dtype=torch.float32)
    task = core.EigenFunctionTask(Traj(traj, w, 1.0), _position_layer(n_atoms, ref, dev), model, str(tmp_path), 12.0, [1.0, 0.5],
                                  diag_coeff=a, beta=1.0, lag_tau=0, learning_rate=2e-3, k=k, batch_size=100, num_epochs=3,
                                  device=dev, verbose=False, save_model_every_step=0)
    assert task._flat._views is not None and task._flat.desc.dims[2] == 20          # padded 10 -> 20
    assert {n: tuple(p.shape) for n, p in model.state_dict().items()} == {n: tuple(p.shape) for n, p in sd0.items()}
    loss, eig, npl, pen, cvec = task.loss_func(torch.tensor(traj), torch.tensor(w), None, None)
    task.backward()
    torch.set_default_dtype(torch.float64)
    sd = {n: p.double().requires_grad_(True) for n, p in sd0.items()}
    feats = [("position", tuple(range(n_atoms)))]
    X = torch.tensor(traj, dtype=torch.float64, requires_grad=True)
    lo, eo, no, po, co = losses.ef_loss(sd, k, AlignFeature(list(range(n_atoms)), ref, feats, False), X, torch.tensor(w), alpha=12.0,
                                        eig_w=[1.0, 0.5], diag_coeff=a.double(), beta=1.0)
    lo.backward()
    torch.set_default_dtype(torch.float32)
    np.testing.assert_allclose(float(loss), float(lo.detach()), rtol=2e-5)
    np.testing.assert_allclose(eig.numpy(), eo.numpy(), rtol=2e-5)
    assert list(cvec) == list(co)
    want = torch.cat([sd[n].grad.reshape(-1) for n, _ in model.named_parameters()]).numpy()
    got = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=4e-4, atol=4e-4 * np.abs(want).max())
    # training keeps the padding at exactly zero
    np.random.seed(2)
    task.train()
    real = torch.zeros_like(task._flat.theta, dtype=torch.bool)
    for p, _ in task._flat.grad_views():
        pass
    mask = torch.ones_like(task._flat.theta)
    saved = {n: p.detach().clone() for n, p in model.named_parameters()}
    for p in model.parameters():
        p.data.zero_()                                  # zero the real entries through the views ...
    assert float(task._flat.theta.abs().max()) == 0.0   # ... and nothing is left: the padding never moved
    for n, p in model.named_parameters():
        p.data.copy_(saved[n])
    assert np.isfinite(np.stack([e[0].numpy() for e in task.loss_list])).all()


def test_comm_abi_single_rank_roundtrip(dev):
    """cvf_comm_* (the data-parallel step's two sums behind the C ABI: RCCL located at run time): id, communicator of one rank,
    in-place fp64 / fp32 all-reduce on the current stream (identity with one rank), destroy.  More ranks need more GPUs than
    the test box has; the multi-rank arithmetic of the step itself is covered by tests/test_dist_gloo.py and tools/check_dp2.py."""
    import ctypes
    from colvarsfinder import _hip
    lib = _hip.lib()
    n = lib.cvf_comm_unique_id_bytes()
    assert n == 128
    uid = torch.zeros(n, dtype=torch.uint8)
    _hip.check(lib.cvf_comm_unique_id(uid.data_ptr()), "cvf_comm_unique_id")
    assert int(uid.to(torch.int64).abs().sum()) > 0
    with torch.cuda.device(dev):
        comm = ctypes.c_void_p()
        _hip.check(lib.cvf_comm_init(ctypes.byref(comm), 0, 1, uid.data_ptr()), "cvf_comm_init")
        a = torch.arange(13, device=dev, dtype=torch.float64) * 0.37
        b = torch.randn(6603, device=dev, dtype=torch.float32)
        a0, b0 = a.clone(), b.clone()
        _hip.check(lib.cvf_comm_allreduce_f64(comm, _hip.ptr(a), a.numel(), _hip.stream()), "cvf_comm_allreduce_f64")
        _hip.check(lib.cvf_comm_allreduce_f32(comm, _hip.ptr(b), b.numel(), _hip.stream()), "cvf_comm_allreduce_f32")
        torch.cuda.synchronize()
        assert torch.equal(a, a0) and torch.equal(b, b0)
        assert lib.cvf_comm_allreduce_f64(None, _hip.ptr(a), 13, _hip.stream()) != 0        # bad argument: error code, not a crash
        assert b"bad argument" in lib.cvf_last_error()
        _hip.check(lib.cvf_comm_destroy(comm), "cvf_comm_destroy")


def test_epoch_graphs_in_chunks_equal_eager_launches(dev, tmp_path):
    """An epoch of more steps than one hipGraph takes (small batches: 300 steps > the 256-step chunk) replays from several graphs
    and gives the numbers of the same training with eager launches, step by step."""
    from colvarsfinder import core, nn
    traj, w, ref = make_molecule_traj(10, 750, seed=77)

    def run(graphs):
        torch.manual_seed(3)
        np.random.seed(3)
        model = nn.EigenFunctions([30, 12, 12, 1], 2)
        task = core.EigenFunctionTask(Traj(traj, w, 1.0), _position_layer(10, ref, dev), model, str(tmp_path), 10.0, [1.0, 0.5], k=2,
                                      batch_size=2, num_epochs=3, learning_rate=1e-3, device=dev, verbose=False, save_model_every_step=0)
        task._use_graphs = graphs
        task.train()
        if graphs:
            assert len(task._graphs) >= 2          # the epoch was captured as more than one graph
        return np.stack([e[0].numpy() for e in task.loss_list]), np.stack([e[1].numpy() for e in task.loss_list])

    tr_g, te_g = run(True)
    tr_e, te_e = run(False)
    assert tr_g.shape[1] == 300 and te_g.shape[1] == 75
    np.testing.assert_array_equal(tr_g, tr_e)
    np.testing.assert_array_equal(te_g, te_e)


def test_colvar_model_is_differentiable_like_the_reference(dev):
    """core.py:372-382, 640-647: the reference's colvar_model() is a plain differentiable Sequential.  A call whose input requires
    grad returns a result WITH an autograd graph (torch operators on the GPU); d xi / d x agrees with autograd through the fp64
    oracle (the alignment through linalg.svd, as the reference differentiates it at core.py:424), for both tasks' models."""
    from colvarsfinder import core, nn, pp
    from oracle import nnref
    from oracle.pp import AlignFeature
    from tests.synth import Traj, diag_coeff_for, make_molecule_traj
    n_atoms, k = 10, 2
    traj, w, ref = make_molecule_traj(n_atoms, 64, seed=41)
    feats = [("position", (0, 2, 3, 5)), ("bond", (0, 1)), ("angle", (1, 2, 3)), ("dihedral", (4, 5, 6, 7))]
    align = [0, 1, 2, 4, 5, 8]
    layer = pp.AlignFeatureLayer(n_atoms, align, ref[align], feats)
    dims = [layer.d_r, 12, 12, 1]
    sd0 = nnref.init_eigenfunctions(dims, k, torch.Generator().manual_seed(2))
    model = nn.EigenFunctions(dims, k)
    model.load_state_dict(sd0)
    a = torch.tensor(diag_coeff_for(n_atoms, 1), dtype=torch.float32)
    task = core.EigenFunctionTask(Traj(traj, w, 1.0), layer, model, "/tmp/cvf_test", 10.0, [1.0, 0.5], diag_coeff=a, k=k, device=dev,
                                  verbose=False, save_model_every_step=0)
    cv = task.colvar_model()
    x = torch.tensor(traj[:16], dtype=torch.float64, requires_grad=True)
    y = cv(x)
    assert y.requires_grad and y.dtype == torch.float64 and y.device == x.device
    (gx,) = torch.autograd.grad(y[:, 1].sum(), x)
    torch.set_default_dtype(torch.float64)
    try:
        xo = torch.tensor(traj[:16], dtype=torch.float64, requires_grad=True)
        yo = nnref.eigenfunctions_forward({n: p.double() for n, p in sd0.items()}, k, AlignFeature(align, ref[align], feats)(xo))
        (go,) = torch.autograd.grad(yo[:, 1].sum(), xo)
    finally:
        torch.set_default_dtype(torch.float32)
    np.testing.assert_allclose(y.detach().numpy(), yo.detach().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(gx.numpy(), go.numpy(), rtol=0, atol=2e-5 * float(go.abs().max()))
    # the kernel path (no grad wanted) gives the same values in fp32
    y32 = cv(torch.tensor(traj[:16])).detach()
    np.testing.assert_allclose(y32.numpy(), yo.detach().numpy(), rtol=2e-5, atol=2e-6)


def test_autoencoder_task_takes_any_module_as_pp_layer(dev):
    """core.py:65,122,635: pp_layer is ANY torch module; AutoEncoderTask applies it once to build the feature trajectory.  A module
    this package has no kernel for (here: pairwise distances, hand-written) is run with torch on the device at construction; the
    training step on the resident feature rows is the usual kernel - first-step loss and a short training against the oracle."""
    from colvarsfinder import core, nn
    from oracle import nnref, train as otrain
    from tests.synth import Traj, make_molecule_traj

    class PairDistances(torch.nn.Module):
        def __init__(self, n_atoms):
            super().__init__()
            i, j = torch.triu_indices(n_atoms, n_atoms, offset=1)
            self.register_buffer("i", i)
            self.register_buffer("j", j)

        def forward(self, x):
            return (x[:, self.i, :] - x[:, self.j, :]).norm(dim=2)

    n_atoms = 6
    traj, w, _ = make_molecule_traj(n_atoms, 400, seed=17)
    layer = PairDistances(n_atoms)
    d_r = n_atoms * (n_atoms - 1) // 2
    e_dims, d_dims = [d_r, 12, 2], [2, 12, d_r]
    sd0 = nnref.init_autoencoder(e_dims, d_dims, torch.Generator().manual_seed(4))
    model = nn.AutoEncoder(e_dims, d_dims)
    model.load_state_dict(sd0)
    want_feat = layer(torch.tensor(traj)).numpy()             # (on the CPU, before the task moves the module to its device)
    task = core.AutoEncoderTask(Traj(traj, w, 1.0), layer, model, "/tmp/cvf_test", learning_rate=2e-3, batch_size=100, num_epochs=2,
                                device=dev, verbose=False, save_model_every_step=0)
    np.testing.assert_allclose(task._feature_traj.cpu().numpy(), want_feat, rtol=1e-6, atol=1e-6)
    np.random.seed(7)
    task.train()
    np.random.seed(7)
    torch.set_default_dtype(torch.float64)
    try:
        res = otrain.train_ae({n: p.double() for n, p in sd0.items()}, PairDistances(n_atoms), traj.astype(np.float64), w, learning_rate=2e-3,
                              batch_size=100, num_epochs=2)
    finally:
        torch.set_default_dtype(torch.float32)
    np.testing.assert_allclose(np.stack([e[0].numpy() for e in task.loss_list]), np.stack([e[0].numpy() for e in res["loss_list"]]), rtol=2e-5)
    cv = task.colvar_model()(torch.tensor(traj[:8]))          # Sequential(pp_layer, encoder) on CPU input, like the notebooks
    assert cv.shape == (8, 2) and torch.isfinite(cv).all()
