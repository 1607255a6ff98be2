"""Oracle (CPU, plain PyTorch / NumPy) for the alignment + feature layer ``r(x)``.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  **Parity unpinned**: the
arithmetic restated here lives in the third-party package ``molann`` (un-vendored,
un-pinned, not installed); the reference only sees it as an opaque
``pp_layer: torch.nn.Module`` (``colvarsfinder/core.py:65,122,403,414,635``) built at
``examples/dipeptide/main.ipynb:333-348``:

    Feature('p1', 'position', input_ag)            main.ipynb:335
    ann.FeatureLayer(feature_list, input_ag)       main.ipynb:337
    ann.AlignmentLayer(align_atom_group, input_ag) main.ipynb:345
    ann.PreprocessingANN(align, feature_mapper)    main.ipynb:348

Algorithm (Kabsch 1976/78):
    c      = unweighted centroid of the align atoms of the frame
    H      = (x_align - c)^T @ ref_c                 [3,3], ref_c = centred reference
    U,S,Vh = svd(H);  d = sign(det(U @ Vh))          (d is a constant for autograd)
    R      = U @ diag(1,1,d) @ Vh
    x_al   = (x - c) @ R                             for every input atom
    features of x_al, concatenated in list order:
        position -> the 3*n coordinates of the listed atoms, atom-major
        bond     -> |x_b - x_a|
        angle    -> cos of the angle at the middle atom (or its value)
        dihedral -> (cos phi, sin phi) (or the value phi = atan2(sin, cos))

``kabsch_vjp_np`` / ``kabsch_jvp_np`` restate, in fp64 NumPy and in closed form, the
derivative of that map that the reference obtains by autograd through
``torch.linalg.svd`` (``core.py:424``); they are the checker for the HIP kernels'
analytic K2/K3 path and are themselves checked against autograd and finite
differences in ``tests/test_oracle_pp.py``.
"""

import numpy as np
import torch

# molann's numeric ids for feature types (the dipeptide notebook prints
# "position ... type_id 3" at main.ipynb:306-307; the others follow molann's
# alphabetical table).
TYPE_ID = {"angle": 0, "bond": 1, "dihedral": 2, "position": 3}
TYPE_NATOMS = {"angle": 3, "bond": 2, "dihedral": 4}


def feature_dim(features, use_angle_value=False):
    d = 0
    for ftype, atoms in features:
        if ftype == "position":
            d += 3 * len(atoms)
        elif ftype == "dihedral":
            d += 1 if use_angle_value else 2
        else:
            d += 1
    return d


def kabsch_rotation(x, align_idx, ref_c, w=None):
    """x [B,N,3], align_idx [n_al] long, ref_c [n_al,3] (centred).  Returns (R [B,3,3], c [B,1,3]).
    ``w`` [n_al] (sum 1): per-atom weights - c is the weighted centroid, H the weighted covariance, ``ref_c`` centred by
    the weighted centroid; ``None`` is molann's uniform alignment."""
    xa = x[:, align_idx, :]
    if w is None:
        c = xa.mean(dim=1, keepdim=True)
        H = torch.matmul((xa - c).transpose(1, 2), ref_c)  # [B,3,3]
    else:
        c = (w[None, :, None] * xa).sum(dim=1, keepdim=True)
        H = torch.matmul((xa - c).transpose(1, 2), w[:, None] * ref_c)
    U, S, Vh = torch.linalg.svd(H)
    d = torch.sign(torch.linalg.det(torch.matmul(U, Vh))).detach()
    D = torch.diag_embed(torch.stack([torch.ones_like(d), torch.ones_like(d), d], dim=1))
    R = torch.matmul(torch.matmul(U, D), Vh)
    return R, c


def _features_grouped(xal, features, use_angle_value):
    """features_of with the features of one type evaluated together (same formulas on [B, n_type, 3] operands), the columns
    put back into list order at the end; the atoms the features read are gathered once."""
    used = sorted({int(i) for _, atoms in features for i in atoms})
    slot = {a: s_ for s_, a in enumerate(used)}
    xs = xal[:, used, :]
    B = xal.shape[0]
    cols, blocks, start = {}, [], 0      # cols[feature number] = its column indices in the concatenated blocks
    by_type = {}
    for n_, (ftype, atoms) in enumerate(features):
        by_type.setdefault(ftype, []).append((n_, [slot[int(i)] for i in atoms]))

    def pick(rows, j):
        return xs[:, [r[1][j] for r in rows], :]

    for ftype, rows in by_type.items():
        if ftype == "position":
            for n_, sl in rows:
                blocks.append(xs[:, sl, :].reshape(B, -1))
                cols[n_] = list(range(start, start + 3 * len(sl)))
                start += 3 * len(sl)
            continue
        if ftype == "bond":
            val = torch.linalg.norm(pick(rows, 1) - pick(rows, 0), dim=2)
            width = 1
        elif ftype == "angle":
            r21, r23 = pick(rows, 0) - pick(rows, 1), pick(rows, 2) - pick(rows, 1)
            cs = (r21 * r23).sum(2) / (torch.linalg.norm(r21, dim=2) * torch.linalg.norm(r23, dim=2))
            val = torch.acos(cs) if use_angle_value else cs
            width = 1
        elif ftype == "dihedral":
            r12, r23, r34 = pick(rows, 1) - pick(rows, 0), pick(rows, 2) - pick(rows, 1), pick(rows, 3) - pick(rows, 2)
            n1, n2 = torch.linalg.cross(r12, r23, dim=2), torch.linalg.cross(r23, r34, dim=2)
            inv = 1.0 / (torch.linalg.norm(n1, dim=2) * torch.linalg.norm(n2, dim=2))
            cs = (n1 * n2).sum(2) * inv
            sn = (n1 * r34).sum(2) * torch.linalg.norm(r23, dim=2) * inv
            if use_angle_value:
                val, width = torch.atan2(sn, cs), 1
            else:
                val, width = torch.stack([cs, sn], dim=2).reshape(B, -1), 2
        else:
            raise ValueError(ftype)
        blocks.append(val)
        for j, (n_, _) in enumerate(rows):
            cols[n_] = list(range(start + width * j, start + width * (j + 1)))
        start += width * len(rows)
    order = [c for n_ in range(len(features)) for c in cols[n_]]
    return torch.cat(blocks, dim=1)[:, order]


def features_of(xal, features, use_angle_value=False, compact=False):
    """xal [B,N,3] -> [B,d_r]; list order, molann conventions (see module docstring).
    ``compact``: the same formulas evaluated per feature TYPE on stacked operands, the atoms gathered once - the autograd graph
    of a 5000-atom frame with ~200 features is then a few dozen nodes instead of thousands (oracle/chunked.py); equal to the
    per-feature evaluation to rounding (tests/test_oracle_pp.py)."""
    if compact:
        return _features_grouped(xal, features, use_angle_value)
    out = []
    for ftype, atoms in features:
        if ftype == "position":
            out.append(xal[:, list(atoms), :].reshape(xal.shape[0], -1))
        elif ftype == "bond":
            a, b = atoms
            out.append(torch.linalg.norm(xal[:, b] - xal[:, a], dim=1, keepdim=True))
        elif ftype == "angle":
            a, b, c = atoms
            r21 = xal[:, a] - xal[:, b]
            r23 = xal[:, c] - xal[:, b]
            cs = (r21 * r23).sum(1, keepdim=True) / (
                torch.linalg.norm(r21, dim=1, keepdim=True) * torch.linalg.norm(r23, dim=1, keepdim=True))
            out.append(torch.acos(cs) if use_angle_value else cs)
        elif ftype == "dihedral":
            a, b, c, d = atoms
            r12 = xal[:, b] - xal[:, a]
            r23 = xal[:, c] - xal[:, b]
            r34 = xal[:, d] - xal[:, c]
            n1 = torch.linalg.cross(r12, r23)
            n2 = torch.linalg.cross(r23, r34)
            inv = 1.0 / (torch.linalg.norm(n1, dim=1, keepdim=True) * torch.linalg.norm(n2, dim=1, keepdim=True))
            cs = (n1 * n2).sum(1, keepdim=True) * inv
            sn = (n1 * r34).sum(1, keepdim=True) * torch.linalg.norm(r23, dim=1, keepdim=True) * inv
            if use_angle_value:
                out.append(torch.atan2(sn, cs))
            else:
                out.append(cs)
                out.append(sn)
        else:
            raise ValueError(ftype)
    return torch.cat(out, dim=1)


class AlignFeature(torch.nn.Module):
    """Oracle twin of molann's ``PreprocessingANN(AlignmentLayer, FeatureLayer)``.

    ``align_idx``: local (0-based, into the input atoms) indices of the align atoms.
    ``ref_pos``: [n_al,3] positions of those atoms in the reference frame; stored minus
    their unweighted centroid (what ``align.show_info()`` prints, main.ipynb:316-327).
    ``features``: list of ``(type_name, atom_index_tuple)``.
    ``align_weights``: optional [n_al] per-atom weights (the north star's "weighted Kabsch"; molann itself aligns with
    uniform weights): weighted centroids, weighted covariance.
    """

    def __init__(self, align_idx, ref_pos, features, use_angle_value=False, align_weights=None, compact=False):
        super().__init__()
        self.compact = bool(compact)
        ref = torch.as_tensor(np.asarray(ref_pos), dtype=torch.get_default_dtype())
        self.register_buffer("align_idx", torch.as_tensor(np.asarray(align_idx), dtype=torch.long))
        if align_weights is None:
            self.w = None
            self.register_buffer("ref_c", ref - ref.mean(dim=0, keepdim=True))
        else:
            w = torch.as_tensor(np.asarray(align_weights, dtype=np.float64), dtype=torch.get_default_dtype())
            self.register_buffer("w", w / w.sum())
            self.register_buffer("ref_c", ref - (self.w[:, None] * ref).sum(dim=0, keepdim=True))
        self.features = [(t, tuple(int(i) for i in a)) for t, a in features]
        self.use_angle_value = bool(use_angle_value)

    def align(self, x):
        R, c = kabsch_rotation(x, self.align_idx, self.ref_c.to(x.dtype), None if self.w is None else self.w.to(x.dtype))
        return torch.matmul(x - c, R)

    def forward(self, x):
        return features_of(self.align(x), self.features, self.use_angle_value, self.compact)


# --------------------------------------------------------------------------------------
# Closed-form derivative of the alignment (fp64 NumPy), position features only.
# Invariant features (bond/angle/dihedral) do not see the alignment at all: their
# derivative w.r.t. the raw coordinates is the plain derivative of the feature.
# --------------------------------------------------------------------------------------

def _ax(T):
    """axial vector of T - T^T:  sum_ij T_ij [s]x_ij == s . _ax(T)."""
    return np.array([T[2, 1] - T[1, 2], T[0, 2] - T[2, 0], T[1, 0] - T[0, 1]])


def _skew(s):
    return np.array([[0.0, -s[2], s[1]], [s[2], 0.0, -s[0]], [-s[1], s[0], 0.0]])


def kabsch_frame_np(x, align_idx, ref_c):
    """One frame, fp64.  Returns (R, c, Kinv) with K = tr(P) I - P, P = R^T H (symmetric)."""
    x = np.asarray(x, dtype=np.float64)
    ref_c = np.asarray(ref_c, dtype=np.float64)
    xa = x[align_idx]
    c = xa.mean(axis=0)
    H = (xa - c).T @ ref_c
    U, S, Vh = np.linalg.svd(H)
    d = np.sign(np.linalg.det(U @ Vh))
    R = U @ np.diag([1.0, 1.0, d]) @ Vh
    P = R.T @ H
    K = np.trace(P) * np.eye(3) - 0.5 * (P + P.T)
    return R, c, np.linalg.inv(K)


def kabsch_vjp_np(x, align_idx, ref_c, gpos, pos_atoms):
    """Gradient w.r.t. raw x [N,3] of  sum_a gpos[a] . ((x[pos_atoms[a]] - c) @ R)."""
    x = np.asarray(x, dtype=np.float64)
    ref_c = np.asarray(ref_c, dtype=np.float64)
    R, c, Kinv = kabsch_frame_np(x, align_idx, ref_c)
    n_al = len(align_idx)
    G = np.zeros_like(x)
    p = gpos @ R.T                      # p_a = R g_a
    M = np.zeros((3, 3))
    for a, ia in enumerate(pos_atoms):
        G[ia] += p[a]
        M += np.outer(x[ia] - c, gpos[a])
    G[align_idx] -= p.sum(axis=0) / n_al
    s = Kinv @ _ax(R.T @ M)
    Z = R @ _skew(s)
    zr = ref_c @ Z.T                    # (Z ref_b) per align atom
    G[align_idx] += zr - zr.mean(axis=0)
    return G


def kabsch_jvp_np(x, align_idx, ref_c, u, pos_atoms):
    """Directional derivative of the aligned positions of ``pos_atoms`` along u [N,3]."""
    x = np.asarray(x, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    ref_c = np.asarray(ref_c, dtype=np.float64)
    R, c, Kinv = kabsch_frame_np(x, align_idx, ref_c)
    ub = u[align_idx].mean(axis=0)
    dH = (u[align_idx] - ub).T @ ref_c
    w = Kinv @ _ax(R.T @ dH)
    dR = R @ _skew(w)
    return (u[list(pos_atoms)] - ub) @ R + (x[list(pos_atoms)] - c) @ dR
