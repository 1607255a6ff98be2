"""CPU: pins for the oracle's alignment + feature layer (oracle/pp.py, "parity unpinned":
the arithmetic is third-party molann, absent here - see oracle/__init__.py).

Invariance, known-answer and finite-difference tests (SURVEY.md section 4 ii-iii), and the
closed-form Kabsch derivative (what the HIP kernels implement) against autograd through
torch.linalg.svd (what the reference executes, core.py:424).
"""

import numpy as np
import pytest
import torch

from oracle import pp as opp
from tests.synth import make_molecule_traj, random_rotations

MIXED = [("position", (0, 2, 3, 5)), ("bond", (0, 1)), ("bond", (2, 7)), ("angle", (1, 2, 3)),
         ("dihedral", (0, 1, 2, 3)), ("dihedral", (4, 5, 6, 7)), ("angle", (6, 8, 9))]


@pytest.fixture(autouse=True)
def _f64():
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(torch.float32)


def layer(n, ref, feats=None, align=None, angle_value=False):
    align = list(range(n)) if align is None else align
    feats = [("position", tuple(range(n)))] if feats is None else feats
    return opp.AlignFeature(align, ref[align], feats, angle_value)


def test_reference_is_centred_like_show_info():
    # main.ipynb:316-327 prints reference positions whose columns sum to zero
    _, _, ref = make_molecule_traj(10, 4, 1)
    L = layer(10, ref)
    assert np.abs(L.ref_c.numpy().sum(0)).max() < 1e-12
    assert L.ref_c.shape == (10, 3)


def test_frame_equal_to_reference_gives_identity():
    _, _, ref = make_molecule_traj(12, 4, 2)
    L = layer(12, ref)
    x = torch.tensor(ref[None] + np.array([1.0, -2.0, 0.5]))
    R, c = opp.kabsch_rotation(x, L.align_idx, L.ref_c)
    np.testing.assert_allclose(R[0].numpy(), np.eye(3), atol=1e-12)
    np.testing.assert_allclose(L(x).numpy().reshape(12, 3), L.ref_c.numpy(), atol=1e-12)


@pytest.mark.parametrize("feats,angle_value", [(None, False), (MIXED, False), (MIXED, True)])
def test_rigid_motion_invariance(feats, angle_value):
    traj, _, ref = make_molecule_traj(10, 16, 3, dtype=np.float64)
    L = layer(10, ref, feats, align=[0, 1, 2, 4, 5, 8] if feats else None, angle_value=angle_value)
    rs = np.random.RandomState(0)
    Q = random_rotations(rs, 16)
    t = rs.normal(size=(16, 1, 3))
    moved = np.einsum("bij,baj->bai", Q, traj) + t
    np.testing.assert_allclose(L(torch.tensor(moved)).numpy(), L(torch.tensor(traj)).numpy(), atol=1e-10)


def test_reflection_case_uses_det_fix():
    # a mirrored molecule cannot be superposed by a proper rotation: R must still have det +1
    traj, _, ref = make_molecule_traj(8, 6, 4, dtype=np.float64)
    L = layer(8, ref)
    mirrored = torch.tensor(traj * np.array([1.0, 1.0, -1.0]))
    R, _ = opp.kabsch_rotation(mirrored, L.align_idx, L.ref_c)
    np.testing.assert_allclose(torch.linalg.det(R).numpy(), 1.0, atol=1e-12)
    np.testing.assert_allclose(torch.matmul(R, R.transpose(1, 2)).numpy(), np.broadcast_to(np.eye(3), (6, 3, 3)), atol=1e-12)


def test_rotation_is_optimal():
    traj, _, ref = make_molecule_traj(9, 5, 5, dtype=np.float64)
    L = layer(9, ref)
    x = torch.tensor(traj)
    base = ((L.align(x) - L.ref_c) ** 2).sum(dim=(1, 2))
    rs = np.random.RandomState(1)
    for _ in range(20):
        # small random rotation about the centroid must not decrease the residual
        w = rs.normal(size=3) * 0.05
        Wm = torch.tensor(opp._skew(w))
        Rp = torch.linalg.matrix_exp(Wm)
        pert = ((torch.matmul(L.align(x), Rp) - L.ref_c) ** 2).sum(dim=(1, 2))
        assert (pert >= base - 1e-10).all()


def test_kabsch_rotation_equals_an_independent_published_implementation():
    """The oracle's rotation against SciPy's Rotation.align_vectors (Kabsch / Wahba by SVD, an implementation nobody here wrote):
    the same proper rotation for generic, for nearly planar and for mirrored frames.  (Not a pin in the contract's sense - the
    third-party layer the reference calls is absent - but a second, published statement of the algorithm SURVEY 8 a15 names.)"""
    from scipy.spatial.transform import Rotation
    for n, seed, squash, mirror in ((9, 11, 1.0, False), (40, 12, 1.0, False), (12, 13, 1e-3, False), (10, 14, 1.0, True)):
        traj, _, ref = make_molecule_traj(n, 7, seed, dtype=np.float64)
        ref = ref * np.array([1.0, 1.0, squash])
        traj = traj * np.array([1.0, 1.0, squash])
        if mirror:
            traj = traj * np.array([1.0, 1.0, -1.0])
        L = layer(n, ref)
        R, c = opp.kabsch_rotation(torch.tensor(traj), L.align_idx, L.ref_c)
        for b in range(traj.shape[0]):
            # x_al = (x - c) R as row vectors: the rotation that takes the centred frame onto the reference is R^T as a matrix on columns
            rot, _ = Rotation.align_vectors(L.ref_c.numpy(), traj[b] - c[b].numpy())
            np.testing.assert_allclose(R[b].numpy().T, rot.as_matrix(), atol=5e-9 if squash == 1.0 else 5e-6)


def test_feature_known_answers():
    x = torch.tensor([[[0.0, 0, 0], [1.0, 0, 0], [1.0, 1.0, 0], [1.0, 1.0, 1.0], [2.0, 1.0, 1.0]]])
    f = opp.features_of(x, [("bond", (0, 1)), ("angle", (0, 1, 2)), ("dihedral", (0, 1, 2, 3)), ("dihedral", (1, 2, 3, 4))])
    # bond 1; right angle -> cos 0; dihedral 0-1-2-3 = +90 deg with this sign convention
    np.testing.assert_allclose(f[0, 0].item(), 1.0, atol=1e-14)
    np.testing.assert_allclose(f[0, 1].item(), 0.0, atol=1e-14)
    np.testing.assert_allclose(f[0, 2:4].numpy(), [0.0, 1.0], atol=1e-14)
    fv = opp.features_of(x, [("angle", (0, 1, 2)), ("dihedral", (0, 1, 2, 3))], use_angle_value=True)
    np.testing.assert_allclose(fv[0].numpy(), [np.pi / 2, np.pi / 2], atol=1e-14)
    assert opp.feature_dim(MIXED) == 20 and opp.feature_dim(MIXED, True) == 18


def test_feature_order_is_atom_major_xyz():
    traj, _, ref = make_molecule_traj(6, 3, 6, dtype=np.float64)
    L = layer(6, ref, [("position", (4, 1))])
    out = L(torch.tensor(traj)).numpy()
    al = L.align(torch.tensor(traj)).numpy()
    np.testing.assert_allclose(out, np.concatenate([al[:, 4], al[:, 1]], axis=1), atol=0)


@pytest.mark.parametrize("align", [None, [0, 1, 2, 4, 5, 8]])
def test_closed_form_vjp_matches_autograd_and_fd(align):
    traj, _, ref = make_molecule_traj(10, 4, 7, dtype=np.float64)
    pos_atoms = [0, 2, 3, 5, 9]
    L = layer(10, ref, [("position", tuple(pos_atoms))], align=align)
    rs = np.random.RandomState(2)
    for b in range(4):
        g = rs.normal(size=(len(pos_atoms), 3))
        x = torch.tensor(traj[b:b + 1], requires_grad=True)
        (L(x) * torch.tensor(g.reshape(1, -1))).sum().backward()
        G = opp.kabsch_vjp_np(traj[b], L.align_idx.numpy(), L.ref_c.numpy(), g, pos_atoms)
        np.testing.assert_allclose(G, x.grad[0].numpy(), rtol=1e-9, atol=1e-11)
        # finite differences of the map itself along a random direction
        u = rs.normal(size=(10, 3))
        eps = 1e-6
        fp = L(torch.tensor(traj[b:b + 1] + eps * u)).detach().numpy()
        fm = L(torch.tensor(traj[b:b + 1] - eps * u)).detach().numpy()
        jv_fd = ((fp - fm) / (2 * eps)).reshape(len(pos_atoms), 3)
        jv = opp.kabsch_jvp_np(traj[b], L.align_idx.numpy(), L.ref_c.numpy(), u, pos_atoms)
        np.testing.assert_allclose(jv, jv_fd, rtol=1e-6, atol=1e-8)
        # adjointness  <J u, g> == <u, J^T g>
        np.testing.assert_allclose((jv * g).sum(), (u * G).sum(), rtol=1e-10)


def test_closed_form_vjp_reflection_branch():
    traj, _, ref = make_molecule_traj(8, 3, 8, dtype=np.float64)
    mirrored = traj * np.array([1.0, 1.0, -1.0])
    L = layer(8, ref)
    rs = np.random.RandomState(3)
    for b in range(3):
        g = rs.normal(size=(8, 3))
        x = torch.tensor(mirrored[b:b + 1], requires_grad=True)
        (L(x) * torch.tensor(g.reshape(1, -1))).sum().backward()
        G = opp.kabsch_vjp_np(mirrored[b], L.align_idx.numpy(), L.ref_c.numpy(), g, list(range(8)))
        np.testing.assert_allclose(G, x.grad[0].numpy(), rtol=1e-8, atol=1e-10)


# ------------------------------------------------------------------------------------------------
# per-atom alignment weights ("weighted Kabsch", north_star; molann's own layer is the uniform case)
# ------------------------------------------------------------------------------------------------
def _weights(n, seed=0):
    return np.random.RandomState(seed).uniform(0.2, 3.0, size=n)


def test_weighted_alignment_uniform_weights_is_the_unweighted_layer():
    traj, _, ref = make_molecule_traj(10, 12, 21, dtype=np.float64)
    x = torch.tensor(traj)
    a = opp.AlignFeature(list(range(10)), ref, MIXED)(x)
    b = opp.AlignFeature(list(range(10)), ref, MIXED, align_weights=np.full(10, 2.5))(x)
    np.testing.assert_allclose(b.numpy(), a.numpy(), atol=1e-12)


def test_weighted_alignment_minimises_the_weighted_residual():
    """(R, c) must be the minimiser of sum_b w_b |(x_b - c) R - ref_b|^2: checked against brute-force perturbations of
    the rotation and against NumPy's SVD of the weighted covariance."""
    n = 9
    traj, _, ref = make_molecule_traj(n, 6, 22, dtype=np.float64)
    w = _weights(n, 3)
    L = opp.AlignFeature(list(range(n)), ref, [("position", tuple(range(n)))], align_weights=w)
    x = torch.tensor(traj)
    wn = torch.tensor(w / w.sum())
    res = lambda al: (wn[None, :, None] * (al - L.ref_c) ** 2).sum(dim=(1, 2))
    base = res(L.align(x))
    rs = np.random.RandomState(2)
    for _ in range(20):
        Rp = torch.linalg.matrix_exp(torch.tensor(opp._skew(rs.normal(size=3) * 0.05)))
        assert (res(torch.matmul(L.align(x), Rp)) >= base - 1e-10).all()
    # independent restatement in NumPy
    for b in range(traj.shape[0]):
        c = (wn.numpy()[:, None] * traj[b]).sum(0)
        rc = ref - (wn.numpy()[:, None] * ref).sum(0)
        U, S, Vt = np.linalg.svd((traj[b] - c).T @ (wn.numpy()[:, None] * rc))
        D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
        np.testing.assert_allclose(L.align(x)[b].numpy(), (traj[b] - c) @ (U @ D @ Vt), atol=1e-10)
    # the weights matter: the unweighted layer gives different positions
    U0 = opp.AlignFeature(list(range(n)), ref, [("position", tuple(range(n)))])
    assert (L.align(x) - L.ref_c - (U0.align(x) - U0.ref_c)).abs().max() > 1e-3


def test_weighted_alignment_rigid_motion_invariance():
    n = 10
    traj, _, ref = make_molecule_traj(n, 8, 23, dtype=np.float64)
    align = [0, 1, 2, 4, 5, 8]
    L = opp.AlignFeature(align, ref[align], MIXED, align_weights=_weights(len(align), 5))
    Rm = random_rotations(np.random.RandomState(4), 8)
    moved = np.einsum("bij,baj->bai", Rm, traj) + np.random.RandomState(6).normal(size=(8, 1, 3)) * 5
    np.testing.assert_allclose(L(torch.tensor(moved)).numpy(), L(torch.tensor(traj)).numpy(), atol=1e-9)
