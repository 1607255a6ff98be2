"""CPU (-m "not gpu"): host-side logic of the shipped package and the C-ABI surface.  No compute
call is made here (there is no GPU); the library is only loaded and its exports checked."""

import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from tests import goldens  # noqa: E402


@pytest.fixture(scope="module")
def built_lib():
    from colvarsfinder import _hip
    if not os.path.exists(_hip.LIB_PATH):
        subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py")], check=True)
    return _hip


def header_functions():
    text = open(os.path.join(ROOT, "include", "cvf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cvf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_lib):
    declared = header_functions()
    assert len(declared) >= 15
    handle = ctypes.CDLL(built_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/cvf.h but not exported"
    # the ctypes binding covers exactly the declared surface
    assert sorted(built_lib.EXPORTED_SYMBOLS) == declared
    lib = built_lib.lib()
    assert lib.cvf_version() >= 100
    assert lib.cvf_ef_nstats(3, 0) == 1 + 3 + 6 + 3 and lib.cvf_ef_nstats(2, 1) == 1 + 2 + 3 + 1 + 6


def test_struct_layouts_match_the_header(built_lib):
    # sizes computed from include/cvf.h by hand: any drift between the C structs and ctypes breaks every call
    assert ctypes.sizeof(built_lib.PPDesc) == 8 * 4 + 7 * 8 + 2 * 4 + 8 + 2 * 8 + 2 * 4   # (+ mrec, slot_row, n_mrec, n_ref)
    assert ctypes.sizeof(built_lib.MLPDesc) == 4 * (2 + 13 + 12 + 2 * 8 * 12 + 1)
    assert ctypes.sizeof(built_lib.EFCfg) == 4 * 4 + 8 * 3 + 8 * 8
    assert ctypes.sizeof(built_lib.AdamArgs) == 3 * 8 + 4 * 8 + 4 * 8     # theta, m, v | lr, beta1, beta2, eps | step_count, mlp, packed, lr_dev


def test_no_cpu_fallback():
    from colvarsfinder import core, nn
    from tests.synth import Traj, make_2d_traj
    traj, w = make_2d_traj(50, 1)
    with pytest.raises(RuntimeError, match="GPU only|no HIP device"):
        core.EigenFunctionTask(Traj(traj, w, 1.0), torch.nn.Identity(), nn.EigenFunctions([2, 8, 1], 1), "/tmp/x", 1.0, [1.0],
                               device=torch.device("cpu"), verbose=False)
    with pytest.raises(RuntimeError, match="GPU only|no HIP device"):
        core.AutoEncoderTask(Traj(traj, w, 1.0), torch.nn.Identity(), nn.AutoEncoder([2, 4, 1], [1, 4, 2]), "/tmp/x",
                             device=torch.device("cpu"), verbose=False)
    # product code never imports the oracle
    for mod in ("core", "nn", "pp", "_hip", "_dist"):
        src = open(os.path.join(ROOT, "colvars-finder_amd", "colvarsfinder", mod + ".py")).read()
        assert "oracle" not in src.replace("the oracle", "").lower() or "import oracle" not in src


def test_nn_api_matches_reference_structure():
    """colvarsfinder.nn twin vs the fixture produced from the reference's nn.py (keys, counts, names, outputs)."""
    from colvarsfinder import nn
    g = np.load(goldens.GOLDEN + "/nn_structure.npz")
    ef = nn.EigenFunctions([30, 20, 20, 20, 1], 3)
    ae = nn.AutoEncoder([66, 20, 20, 20, 2], [2, 10, 10, 66])
    assert list(ef.state_dict().keys()) == [str(s) for s in g["ef_keys"]]
    assert list(ae.state_dict().keys()) == [str(s) for s in g["ae_keys"]]
    assert sum(p.numel() for p in ef.parameters()) == int(g["ef_nparams"])
    assert sum(p.numel() for p in ae.parameters()) == int(g["ae_nparams"])
    assert [n for n, _ in ef.named_modules()] == [str(s) for s in g["ef_modules"]]
    assert [n for n, _ in ae.named_modules()] == [str(s) for s in g["ae_modules"]]
    ef.load_state_dict(goldens.state_dict(g, "ef/"))
    ae.load_state_dict(goldens.state_dict(g, "ae/"))
    assert [n for n, _ in ef.get_params_of_cv(1)] == [str(s) for s in g["ef_cv1_names"]]
    assert [n for n, _ in ae.get_params_of_cv(1)] == [str(s) for s in g["ae_cv1_names"]]
    np.testing.assert_array_equal(ae.get_params_of_cv(1)[-2][1].detach().numpy(), g["ae_cv1_last_w"])
    assert ae.encoded_dim == int(g["ae_encoded_dim"])
    np.testing.assert_allclose(ef(torch.tensor(g["x30"])).detach().numpy(), g["ef_out"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ae(torch.tensor(g["x66"])).detach().numpy(), g["ae_out"], rtol=1e-6, atol=1e-6)
    with pytest.raises(AssertionError):
        nn.create_sequential_nn([3])
    with pytest.raises(AssertionError):
        nn.EigenFunctions([3, 4, 2], 1)
    with pytest.raises(AssertionError):
        nn.AutoEncoder([3, 2], [3, 3])
    # one shared activation instance, named like the reference (nn.py:55-57)
    seq = nn.create_sequential_nn([4, 3, 3, 1])
    assert seq._modules["activation 1"] is seq._modules["activation 2"]


def test_mlp_layout_offsets_follow_parameters_order():
    from colvarsfinder import nn
    ef = nn.EigenFunctions([6, 4, 4, 1], 2)
    lay = nn.mlp_layout(ef)
    assert lay["n_params"] == 2 * (6 * 4 + 4 + 4 * 4 + 4 + 4 + 1)
    assert lay["nets"][0] == [(0, 24, 6, 4, 1), (28, 44, 4, 4, 1), (48, 52, 4, 1, 0)]
    assert lay["nets"][1][0][0] == 53
    ae = nn.AutoEncoder([6, 4, 2], [2, 3, 6])
    chain = nn.mlp_layout(ae)["nets"][0]
    assert [(c[2], c[3], c[4]) for c in chain] == [(6, 4, 1), (4, 2, 0), (2, 3, 1), (3, 6, 0)]
    # activation codes of include/cvf.h; modules outside the kernels' set are rejected with a clear error
    assert [c[4] for c in nn.mlp_layout(nn.EigenFunctions([6, 4, 1], 1, activation=torch.nn.ReLU()))["nets"][0]] == [nn.ACT_RELU, 0]
    assert [c[4] for c in nn.mlp_layout(nn.AutoEncoder([6, 4, 2], [2, 6], torch.nn.Softplus()))["nets"][0]] == [nn.ACT_SOFTPLUS, 0, 0]
    with pytest.raises(NotImplementedError, match="Tanh, Sigmoid"):
        nn.mlp_layout(nn.EigenFunctions([6, 4, 1], 1, activation=torch.nn.GELU()))
    with pytest.raises(NotImplementedError):
        nn.mlp_layout(nn.AutoEncoder([6, 4, 2], [2, 6], torch.nn.ELU(alpha=0.5)))


def test_regautoencoder_side_by_side_chain_layout():
    """RegAutoEncoderTask's flat buffer (core._RegFlatParams): encoder, then decoder and regulariser nets side by side.
    Evaluating the dense chain it describes must reproduce forward_ae | forward_reg; the mask marks exactly the module's
    parameters; the module's tensors alias their blocks."""
    from colvarsfinder import core, nn
    torch.manual_seed(4)
    model = nn.RegAutoEncoder([7, 5, 3], [3, 4, 6, 7], [3, 2, 5, 1], 2)
    want_in = torch.randn(9, 7)
    want = torch.cat([model.forward_ae(want_in), model.forward_reg(want_in)], dim=1).detach()
    n_real = sum(p.numel() for p in model.parameters())
    flat = core._RegFlatParams(model, torch.device("cpu"))
    d = flat.desc
    assert (d.n_nets, d.n_layers) == (1, 2 + 3)
    assert [d.dims[i] for i in range(6)] == [7, 5, 3, 4 + 2 * 2, 6 + 2 * 5, 7 + 2]
    assert [d.act[i] for i in range(5)] == [1, 0, 1, 1, 0]
    h = want_in
    for l in range(d.n_layers):
        fi, fo = d.dims[l], d.dims[l + 1]
        W = flat.theta[d.w_off[0][l]:d.w_off[0][l] + fo * fi].view(fo, fi)
        b = flat.theta[d.b_off[0][l]:d.b_off[0][l] + fo]
        h = h @ W.T + b
        if d.act[l]:
            h = torch.tanh(h)
    torch.testing.assert_close(h, want, rtol=1e-6, atol=1e-6)
    assert int(flat.mask.sum()) == n_real and set(flat.mask.unique().tolist()) <= {0.0, 1.0}
    assert float(flat.theta[flat.mask == 0].abs().max()) == 0.0          # structural zeros of the block-diagonal layers
    assert flat.n == d.n_params > n_real
    # aliasing: writing through the module changes the flat buffer (and the other way round)
    with torch.no_grad():
        model.reg[1][2].weight.add_(1.0)
    p = model.reg[1][2].weight
    l, r0, c0 = 2 + 1, 6 + 5, 4 + 2          # merged layer 1 (second of decoder/regularisers), block of regulariser 1
    W = flat.theta[d.w_off[0][l]:d.w_off[0][l] + d.dims[l + 1] * d.dims[l]].view(d.dims[l + 1], d.dims[l])
    assert torch.equal(W[r0:r0 + 5, c0:c0 + 2], p.data)
    # frozen encoder: its entries leave the mask
    model2 = nn.RegAutoEncoder([7, 5, 3], [3, 4, 7], [3, 4, 1], 1)
    flat2 = core._RegFlatParams(model2, torch.device("cpu"), freeze_encoder=True)
    n_enc = sum(p.numel() for p in model2.encoder.parameters())
    assert int(flat2.mask.sum()) == sum(p.numel() for p in model2.parameters()) - n_enc
    assert float(flat2.mask[:n_enc].sum()) == 0.0
    with pytest.raises(NotImplementedError):   # decoder and regularisers of different depth cannot run side by side
        core._RegFlatParams(nn.RegAutoEncoder([7, 5, 3], [3, 4, 7], [3, 4, 4, 1], 1), torch.device("cpu"))


def test_split_consumes_numpy_rng_like_sklearn():
    from sklearn.model_selection import train_test_split
    from colvarsfinder.core import _split
    for n, ratio, seed in [(5000, 0.2, 1), (4998, 0.2, 2), (257, 0.33, 3)]:
        np.random.seed(seed)
        tr_ref, te_ref = train_test_split(np.arange(n), test_size=ratio)
        after_ref = np.random.rand()
        np.random.seed(seed)
        tr, te = _split(n, ratio)
        assert np.random.rand() == after_ref          # exactly one permutation drawn
        np.testing.assert_array_equal(tr, tr_ref)
        np.testing.assert_array_equal(te, te_ref)


def test_local_slice_partitions_a_batch():
    from colvarsfinder._dist import local_slice
    for n, w in [(20000, 8), (20000, 3), (7, 8), (64, 1)]:
        cuts = [local_slice(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1


class _AtomGroup:
    def __init__(self, ix, positions=None):
        self.ix = np.asarray(ix)
        self.positions = positions

    def __len__(self):
        return len(self.ix)


def test_molann_style_constructors_build_the_descriptor():
    """Feature / FeatureLayer / AlignmentLayer / PreprocessingANN as called at main.ipynb:333-348."""
    from colvarsfinder import _hip, pp
    glob = [1, 4, 5, 6, 8, 10, 14, 15, 16, 18]              # the notebook's heavy-atom indices (main.ipynb:309-313)
    pos = np.random.RandomState(0).normal(size=(10, 3)).astype(np.float32)
    input_ag = _AtomGroup(glob, pos)
    feats = [pp.Feature("p1", "position", input_ag), pp.Feature("d1", "dihedral", _AtomGroup([4, 6, 8, 14])),
             pp.Feature("b1", "bond", _AtomGroup([1, 18]))]
    fl = pp.FeatureLayer(feats, input_ag)
    info = fl.get_feature_info()
    assert list(info["type_id"]) == [3, 2, 1] and list(info["atom indices"][0]) == [g + 1 for g in glob]
    assert fl.output_dimension() == 30 + 2 + 1
    al = pp.AlignmentLayer(input_ag, input_ag)
    np.testing.assert_allclose(al.ref_c.sum(0), 0.0, atol=1e-6)      # centred like show_info() prints
    layer = pp.PreprocessingANN(al, fl)
    assert layer.d_r == 33 and layer.n_atoms == 10
    rec = layer.rec.numpy()
    assert rec.shape == (12, 6)
    assert (rec[:10, 0] == _hip.FEAT_POSITION).all() and list(rec[:10, 1]) == list(range(10))
    assert list(rec[10]) == [_hip.FEAT_DIHEDRAL, 1, 3, 4, 6, 30] and list(rec[11]) == [_hip.FEAT_BOND, 0, 9, 0, 0, 32]
    with pytest.raises(ValueError):
        pp.Feature("x", "torsion", input_ag)
    with pytest.raises(ValueError):
        pp.FeatureLayer([pp.Feature("b", "bond", _AtomGroup([1, 2]))], input_ag)   # atom 2 not in the input group
    with pytest.raises(RuntimeError):
        layer(torch.zeros(2, 10, 3))                                                # CPU tensor: no fallback


def test_utils_match_reference_files(tmp_path):
    """colvarsfinder.utils twin vs the files the reference's utils.py wrote (fixture utils_2d.npz):
    same RNG consumption, byte-identical traj.txt / output.csv / weights.txt, same filtered trajectory."""
    import contextlib
    import io
    from colvarsfinder import utils
    g = np.load(goldens.GOLDEN + "/utils_2d.npz")

    class Pot:
        dim, beta = 2, 1.5

        def V(self, x):
            return (x[0] ** 2 - 1) ** 2 + 2.0 * x[1] ** 2

        def gradV(self, x):
            return np.array([4 * x[0] * (x[0] ** 2 - 1), 4.0 * x[1]])

    tmp = str(tmp_path)
    with contextlib.redirect_stdout(io.StringIO()):
        np.random.seed(77)
        utils.integrate_sde_overdamped(Pot(), 3000, tmp, pre_steps=50, step_size=0.01, report_interval=10,
                                       report_interval_stdout=1000)
        assert np.random.rand() == float(g["rng_after"])
        utils.calc_weights(os.path.join(tmp, "output.csv"), 1.5, 1.0, traj_weight_filename=os.path.join(tmp, "weights.txt"))
        t = utils.WeightedTrajectory(traj_filename=os.path.join(tmp, "traj.txt"), weight_filename=os.path.join(tmp, "weights.txt"),
                                     min_w=0.2, max_w=3.0, verbose=False)
    assert open(os.path.join(tmp, "traj.txt")).read() == str(g["traj_txt"])
    assert open(os.path.join(tmp, "output.csv")).read() == str(g["csv_txt"])
    assert open(os.path.join(tmp, "weights.txt")).read() == str(g["w_txt"])
    np.testing.assert_array_equal(t.trajectory, g["wt_traj"])
    np.testing.assert_allclose(t.weights, g["wt_weights"], rtol=1e-15)
    assert t.dt == float(g["wt_dt"]) and t.n_frames == int(g["wt_n_frames"])
    with pytest.raises(FileNotFoundError):
        utils.WeightedTrajectory(traj_filename=os.path.join(tmp, "nope.txt"))
    with pytest.raises(NotImplementedError):
        utils.integrate_md_langevin()


def test_slot_record_batches_and_row_tables():
    """pp._batch_records: every record appears once, batches are 64 wide and of one type, padding entries have type -1
    (CVF_PP_SLOT_BATCHED).  The derivative kernel's tables (include/cvf.h: mrec, slot_row): every (record, atom position)
    pair owns exactly one row, the rows of a slot are contiguous and in record order, urow = the slot's first row."""
    from colvarsfinder import _hip, pp
    rs = np.random.RandomState(5)
    natoms = {_hip.FEAT_POSITION: 1, _hip.FEAT_BOND: 2, _hip.FEAT_ANGLE: 3, _hip.FEAT_DIHEDRAL: 4}
    recs, out = [], 0
    for t, n in ((_hip.FEAT_POSITION, 40), (_hip.FEAT_BOND, 150), (_hip.FEAT_ANGLE, 70), (_hip.FEAT_DIHEDRAL, 130)):
        for _ in range(n):
            atoms = [int(a) for a in rs.choice(30, natoms[t], replace=False)]    # few slots: plenty of shared atoms
            recs.append([t] + atoms + [0] * (4 - len(atoms)) + [out])
            out += 3 if t == _hip.FEAT_POSITION else 1
    batched = pp._batch_records(sorted(recs, key=lambda r: r[0]))
    assert len(batched) % 64 == 0
    real = [r for r in batched if r[0] >= 0]
    assert sorted(map(tuple, real)) == sorted(map(tuple, recs))
    for b in range(0, len(batched), 64):
        assert len({r[0] for r in batched[b:b + 64] if r[0] >= 0}) <= 1
    n_at = 400
    names = {_hip.FEAT_POSITION: "position", _hip.FEAT_BOND: "bond", _hip.FEAT_ANGLE: "angle", _hip.FEAT_DIHEDRAL: "dihedral"}
    chain = [("dihedral", (i, i + 1, i + 2, i + 3)) for i in range(0, 300)]       # a backbone: neighbours share atoms
    mixed = [(names[r[0]], tuple(r[1:1 + natoms[r[0]]])) for r in recs]
    for feats in (chain, mixed):
        lay = pp.AlignFeatureLayer(n_at, list(range(n_at)), rs.normal(size=(n_at, 3)), feats)
        assert lay._flags & _hip.PP_SLOT_BATCHED and lay._n_rec_slot % 64 == 0
        mrec, slot_row = lay.mrec.numpy().view(np.uint32), lay.slot_row.numpy()
        n_ref = lay._n_ref
        assert slot_row[0] == 0 and slot_row[-1] == n_ref and np.all(np.diff(slot_row) >= 1) and len(slot_row) == lay._n_slot + 1
        seen = np.zeros(n_ref, dtype=int)
        last_rec_of_row = {}
        for i, m in enumerate(mrec):
            ty = int(m[0] & 7) - 1
            sl = [m[1] & 0xffff, m[1] >> 16, m[2] & 0xffff, m[2] >> 16]
            ur = [m[3] & 0xffff, m[3] >> 16, m[4] & 0xffff, m[4] >> 16]
            rw = [int(ur[j]) + int((m[5] >> (8 * j)) & 255) for j in range(4)]
            for j in range(natoms[ty]):
                seen[rw[j]] += 1
                assert slot_row[sl[j]] <= rw[j] < slot_row[sl[j] + 1] and ur[j] == slot_row[sl[j]]
                last_rec_of_row[int(rw[j])] = i
        assert np.all(seen == 1)
        for t in range(lay._n_slot):      # rows of a slot in record order (a fixed summation order)
            owners = [last_rec_of_row[r] for r in range(slot_row[t], slot_row[t + 1])]
            assert owners == sorted(owners)


@pytest.mark.parametrize("angle_value,weighted", [(False, False), (True, False), (False, True)])
def test_export_twin_matches_oracle_and_scripts(tmp_path, angle_value, weighted):
    """export.ScriptableAlignFeature (the TorchScript export twin of the kernel-backed layer, reference save_model
    core.py:205-226): same features as the oracle layer, scriptable, and the saved file reloads and reproduces them."""
    from colvarsfinder import export, pp
    from oracle.pp import AlignFeature
    from tests.synth import make_molecule_traj
    n_atoms = 14
    traj, _, ref = make_molecule_traj(n_atoms, 40, seed=77, scale=2.0, sigma=0.3)
    feats = [("position", (0, 3, 4, 9)), ("bond", (0, 1)), ("dihedral", (1, 2, 3, 4)), ("angle", (5, 6, 7)), ("bond", (8, 13)),
             ("dihedral", (9, 10, 11, 12))]
    align = [0, 1, 2, 3, 4, 5, 6, 7, 9, 11]
    aw = np.random.RandomState(5).uniform(0.3, 2.0, size=len(align)) if weighted else None
    layer = pp.AlignFeatureLayer(n_atoms, align, ref[align], feats, angle_value, align_weights=aw)
    twin = export.ScriptableAlignFeature(layer)
    x = torch.tensor(traj, dtype=torch.float64)
    want = AlignFeature(align, ref[align], feats, angle_value, align_weights=aw).double()(x).numpy()
    got = twin(x).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)   # the layer keeps its reference in fp32
    cv = export.scriptable_cv(torch.nn.Sequential(layer, torch.nn.Linear(layer.d_r, 2)))
    path = str(tmp_path / "scripted_cv_cpu.pt")
    torch.jit.script(cv).save(path)
    loaded = torch.jit.load(path)
    x32 = torch.tensor(traj, dtype=torch.float32)
    np.testing.assert_allclose(loaded(x32).detach().numpy(), cv(x32).detach().numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("n,bs,world", [(4000, 1000, 2), (3998, 1000, 8), (1000, 1001, 4), (80000, 20000, 8), (317, 100, 3)])
def test_shard_batches_partitions_every_global_batch(n, bs, world):
    """SURVEY 8e partitioning: every rank keeps only its slice of every static batch; the slices of one batch tile that
    batch exactly, in order; the dropped tail (drop_last=True, core.py:474) is resident nowhere; a rank holds ~1/world."""
    from colvarsfinder import _dist
    bs = min(bs, n)
    plans = [_dist.shard_batches(n, bs, r, world) for r in range(world)]
    n_batches = n // bs
    for j in range(n_batches):
        rows = np.concatenate([pos[j * nb:(j + 1) * nb] for pos, nb in plans])
        assert np.array_equal(rows, np.arange(j * bs, (j + 1) * bs))
    total = sum(len(pos) for pos, _ in plans)
    assert total == n_batches * bs
    for pos, nb in plans:
        assert len(pos) == n_batches * nb and abs(nb - bs / world) < 1
    assert _dist.shard_batches(n, bs, 0, 1)[0].tolist() == list(range(n_batches * bs))


@pytest.mark.parametrize("width", [21, 24, 30, 32])
def test_one_wide_hidden_layer_has_no_padded_layout(width):
    """EigenFunctions([d, 21..32, 1]) has no kernel width to pad to (one hidden layer: at most 20 units);
    _FlatParams._init_padded must answer False - the task then raises NotImplementedError - instead of crashing in min()."""
    from colvarsfinder import _hip, core, nn
    assert max(_hip.ef_widths(1)) < width
    flat = object.__new__(core._FlatParams)
    assert core._FlatParams._init_padded(flat, nn.EigenFunctions([6, width, 1], 2), torch.device("cpu")) is False


def test_bench_gpus_n_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` outside a launcher starts `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` as a CHILD process (no exec, nothing
    in the parent touches the GPU) and exits with the child's return code."""
    import bench
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # inside a launcher (RANK set) the same command line does NOT spawn again
    monkeypatch.setenv("RANK", "0")
    seen.clear()
    with pytest.raises(BaseException):      # (no GPU here: it fails later, but not in launch_ranks)
        bench.main()
    assert "cmd" not in seen


def test_row_reader_serves_rows_from_a_memory_mapped_file(tmp_path):
    """utils.RowReader / MappedTrajectory: rows of a memory-mapped .npy trajectory in the order asked for (read in ascending
    file order), with the count of what was read - the host side of shard residency (a rank reads its own rows only)."""
    from colvarsfinder.utils import MappedTrajectory, RowReader
    rs = np.random.RandomState(3)
    x = rs.standard_normal((1000, 7, 3)).astype(np.float32)
    path = str(tmp_path / "traj.npy")
    np.save(path, x)
    t = MappedTrajectory(path, dt=0.25)
    assert t.n_frames == 1000 and t.dt == 0.25 and t.weights.shape == (1000,) and t.trajectory.shape == (1000, 7, 3)
    rows = rs.permutation(1000)[:300]
    got = t.trajectory.take_rows(rows)
    assert got.dtype == np.float32 and np.array_equal(got, x[rows])
    assert t.trajectory.rows_read == 300 and t.trajectory.bytes_read == 300 * 7 * 3 * 4
    assert np.array_equal(t.trajectory[0], x[0])
    assert np.array_equal(RowReader(x).take_rows([5, 5, 1]), x[[5, 5, 1]])     # repeats, any order
    assert RowReader(x).take_rows(np.zeros(0, dtype=np.int64)).shape == (0, 7, 3)
    from colvarsfinder.core import _HostFrames
    for src in (x, np.load(path, mmap_mode="r"), t.trajectory):
        h = _HostFrames(src)
        assert h.shape == (1000, 7, 3) and np.array_equal(h.rows(rows), x[rows]) and np.array_equal(h.all(), x)


def test_derivative_table_limits_are_reported_at_construction():
    """the large-molecule derivative kernel's tables pack row offsets into 8 bits (fewer than 256 features may share one
    atom) and slots / rows into 16 bits; a feature list past a limit is named by AlignFeatureLayer.derivative_table_limits() - which
    EigenFunctionTask's generator mode turns into a NotImplementedError at construction - instead of a generic message at the
    first training step."""
    from colvarsfinder import pp
    n_atoms = 400
    ref = np.random.RandomState(0).normal(size=(n_atoms, 3))
    ok = pp.AlignFeatureLayer(n_atoms, list(range(n_atoms)), ref, [("position", (0, 1, 2))] + [("bond", (3, 4 + i)) for i in range(100)])
    assert ok.derivative_table_limits() is None
    crowded = pp.AlignFeatureLayer(n_atoms, list(range(n_atoms)), ref, [("bond", (3, 4 + i)) for i in range(300)])
    why = crowded.derivative_table_limits()
    assert why is not None and "share one atom" in why and "256" in why
