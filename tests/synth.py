"""Seeded synthetic trajectories shared by the golden generator, the tests and bench.py.

Follows SURVEY.md section 8(d): frame b is a randomly rotated and translated noisy copy
of a random reference structure, ``x_b = Q_b (ref + sigma xi_b) + t_b``; weights are
U(0.2, 2) and mean-normalised like ``WeightedTrajectory`` does (utils.py:145,159).
NumPy only (no torch, no GPU), so the same generator feeds the CPU baseline.
"""

import numpy as np


class Traj:
    """Duck-typed stand-in for ``colvarsfinder.utils.WeightedTrajectory`` - the tasks read
    exactly three attributes (core.py:329,343-346,634-635)."""

    def __init__(self, trajectory, weights, dt):
        self.trajectory = trajectory
        self.weights = weights
        self.dt = dt
        self.n_frames = trajectory.shape[0]


def random_rotations(rs, n):
    q = rs.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - z * w)
    R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w)
    R[:, 2, 1] = 2 * (y * z + x * w)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def make_weights(rs, n):
    w = rs.uniform(0.2, 2.0, size=n)
    return w / w.mean()


def make_molecule_traj(n_atoms, n_frames, seed, scale=2.0, sigma=0.3, dtype=np.float32):
    """Returns (traj [n,N,3] dtype, weights [n] float64, ref [N,3] float64)."""
    rs = np.random.RandomState(seed)
    ref = rs.normal(scale=scale, size=(n_atoms, 3))
    xi = rs.normal(scale=sigma, size=(n_frames, n_atoms, 3))
    Q = random_rotations(rs, n_frames)
    t = rs.normal(size=(n_frames, 1, 3))
    x = np.einsum("bij,baj->bai", Q, ref[None] + xi) + t
    return x.astype(dtype), make_weights(rs, n_frames), ref


def make_2d_traj(n_frames, seed, dtype=np.float64):
    """Two-well cloud in the plane, shape [n,2] like a text-file trajectory (utils.py:135-138)."""
    rs = np.random.RandomState(seed)
    side = rs.choice([-1.0, 1.0], size=n_frames)
    x = np.stack([side + 0.35 * rs.normal(size=n_frames), 0.6 * rs.normal(size=n_frames)], axis=1)
    return x.astype(dtype), make_weights(rs, n_frames)


def diag_coeff_for(n_atoms, seed):
    """'Diffusion-weighted' diagonal a = 1/m_j repeated x3, m_j in {1,12,14,16}."""
    rs = np.random.RandomState(seed)
    m = rs.choice([1.0, 12.0, 14.0, 16.0], size=n_atoms)
    return np.repeat(1.0 / m, 3)
