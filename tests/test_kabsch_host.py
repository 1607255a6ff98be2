"""The alignment solver of csrc/cvf_kabsch.hpp (fp32 Jacobi guess + fp64 Newton polish on the rotation group, fp64 Jacobi
for the frames fp32 cannot resolve) compiled in hipcc's HOST pass (tools/kabsch_host.hip) and run on the CPU against an fp64
SVD - the definition the oracle (oracle/pp.py) and the reference's molann layer use: R = U diag(1, 1, det(U V^T)) V^T.
No GPU needed: this is the very source the kernels inline."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def solver(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = str(tmp_path_factory.mktemp("kabsch") / "kabsch_host.so")
    subprocess.run([HIPCC, "-O2", "-std=c++17", "-shared", "-fPIC", "--cuda-host-only", "--offload-arch=gfx950",
                    "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "colvars-finder_amd", "csrc"), "-Wno-pass-failed",
                    os.path.join(ROOT, "tools", "kabsch_host.hip"), "-o", out], check=True, timeout=300)
    lib = ctypes.CDLL(out)

    def run(H):
        H = np.ascontiguousarray(np.asarray(H, dtype=np.float64).reshape(-1, 9))
        n = len(H)
        R, K, R0 = np.zeros((n, 9), np.float32), np.zeros((n, 6), np.float32), np.zeros((n, 9), np.float32)
        ok = np.zeros(n, np.int32)
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        lib.cvf_test_kabsch(P(H), n, P(R), P(K), P(R0), P(ok))
        Kf = np.stack([K[:, 0], K[:, 1], K[:, 2], K[:, 1], K[:, 3], K[:, 4], K[:, 2], K[:, 4], K[:, 5]], 1).reshape(n, 3, 3)
        Rr = np.zeros((n, 9), np.float32)
        lib.cvf_test_kabsch_rotation_only(P(H), n, P(Rr))
        run.rotation_only = Rr.reshape(n, 3, 3).astype(np.float64)
        return R.reshape(n, 3, 3).astype(np.float64), Kf.astype(np.float64), R0.reshape(n, 3, 3).astype(np.float64), ok.astype(bool)

    return run


def svd_rotation(H):
    U, S, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(U @ Vt))
    D = np.zeros_like(H)
    D[:, 0, 0] = D[:, 1, 1] = 1.0
    D[:, 2, 2] = d
    return U @ D @ Vt, S, d


def covariances(rs, n, kinds):
    out = []
    for t in range(n):
        kind = kinds[t % len(kinds)]
        na = rs.randint(3, 30)
        ref = rs.normal(size=(na, 3)) * rs.uniform(0.5, 5)
        if kind == "near_planar":
            ref[:, 2] *= 1e-3
        if kind == "planar":
            ref[:, 2] = 0
        ref -= ref.mean(0)
        Q, _ = np.linalg.qr(rs.normal(size=(3, 3)))
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        x = (ref + rs.normal(size=(na, 3)) * rs.uniform(0.01, 1.0)) @ Q.T
        if kind == "unrelated":
            x = rs.normal(size=(na, 3))
        if kind == "exact":
            x = ref @ Q.T
        x -= x.mean(0)
        H = x.T @ ref
        if kind == "near_collinear":   # singular values (1, s, <= s / 2) with s = 1e-5 .. 1e-3, either sign of the determinant
            U, _, Vt = np.linalg.svd(H)
            s2 = 10.0 ** rs.uniform(-5, -3)
            H = U @ np.diag([1.0, s2, s2 * rs.uniform(0, 0.5) * rs.choice([-1.0, 1.0])]) @ Vt * rs.uniform(0.1, 100)
        if t % 7 == 0:
            H *= 1e5      # thousands of atoms, Angstrom
        if t % 11 == 0:
            H *= 1e-6     # a handful of atoms, nanometres
        out.append(H)
    return np.array(out)


def test_rotation_and_kinv_match_the_svd(solver):
    rs = np.random.RandomState(2)
    H = covariances(rs, 20000, ["noisy", "near_planar", "planar", "unrelated", "noisy", "exact"])
    R, K, R0, resolved = solver(H)
    Rr, S, d = svd_rotation(H)
    gap = (S[:, 1] + d * S[:, 2]) / S[:, 0]          # the rotation is unique iff sigma_2 + det sigma_3 > 0
    ok = gap > 1e-5
    assert ok.sum() > 19900 and np.isfinite(R).all() and np.isfinite(K).all()
    err = np.abs(R - Rr).max((1, 2))
    assert err[ok].max() < 1e-7, err[ok].max()       # fp32 rounding of the stored rotation; the fp64 value is ~1e-12 off
    # the rotation-only variant (features without K^-1: one Newton step) stores the same rotation up to fp32 rounding
    assert np.abs(solver.rotation_only - Rr).max((1, 2))[ok].max() < 1.5e-7
    # the fp32 stage alone is orders of magnitude off on ill-conditioned frames - the polish is what carries the precision
    assert np.abs(R0 - Rr).max((1, 2))[ok].max() > 1e-4
    # Kinv = (tr(P) I - P)^-1, P = sym(R^T H)
    Pm = np.einsum("nji,njk->nik", Rr, H)
    Sm = 0.5 * (Pm + Pm.transpose(0, 2, 1))
    Km = np.trace(Sm, axis1=1, axis2=2)[:, None, None] * np.eye(3) - Sm
    Ki = np.linalg.inv(Km[ok])
    assert (np.abs(K[ok] - Ki).max((1, 2)) / np.abs(Ki).max((1, 2))).max() < 3e-7
    # proper rotations
    assert np.abs(np.einsum("nji,njk->nik", R, R) - np.eye(3)).max() < 5e-7
    assert np.all(np.linalg.det(R) > 0.999)
    # ordinary molecules do not need the fp64 stage (a random 3-atom set can be nearly collinear: a handful in 20 000)
    assert resolved.mean() > 0.999


def test_frames_fp32_cannot_resolve_take_the_fp64_stage(solver):
    """Nearly collinear align sets: sigma_2 / sigma_1 ~ 1e-5 .. 1e-3, so the second eigenvalue of H^T H sits 1e-10 .. 1e-6 below the
    first - under fp32 resolution.  The fp32 stage reports it and the fp64 stage takes over; the answer still matches the SVD."""
    rs = np.random.RandomState(5)
    H = covariances(rs, 3000, ["near_collinear"])
    R, K, R0, resolved = solver(H)
    Rr, S, d = svd_rotation(H)
    gap = (S[:, 1] + d * S[:, 2]) / S[:, 0]
    ok = gap > 1e-6
    assert ok.sum() > 2000 and (~resolved).sum() > 500
    assert np.abs(R - Rr).max((1, 2))[ok].max() < 2e-7


def test_degenerate_inputs_stay_finite(solver):
    H = np.zeros((6, 3, 3))
    H[1] = np.diag([1.0, 0.0, 0.0])                  # rank 1
    H[2] = np.diag([1.0, 1.0, -1.0])                 # reflection with sigma_2 = sigma_3: not unique
    H[3] = np.eye(3)
    H[4] = 1e-20 * np.eye(3)
    H[5] = np.diag([3.0, 2.0, -1.0])                 # reflection, unique: R = diag(1, 1, -1)-corrected identity
    R, K, R0, resolved = solver(H)
    assert np.isfinite(R).all() and np.isfinite(K).all()
    np.testing.assert_allclose(R[3], np.eye(3), atol=1e-7)
    np.testing.assert_allclose(R[4], np.eye(3), atol=1e-7)
    np.testing.assert_allclose(R[5], np.eye(3), atol=1e-7)
    # (rank <= 1 - every align atom on one line through the centroid - has no second axis: the stored matrix is the rank-one
    #  u1 v1^T, finite; the SVD's answer there is an arbitrary member of a circle of optimal rotations)
    for i in (2, 3, 4, 5):                            # whatever is returned otherwise is a proper rotation
        np.testing.assert_allclose(R[i].T @ R[i], np.eye(3), atol=1e-6)
        assert np.linalg.det(R[i]) > 0.999
