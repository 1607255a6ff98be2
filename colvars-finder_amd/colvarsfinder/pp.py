"""Preprocessing layer ``r(x)``: Kabsch alignment to a reference structure + feature map.

The reference takes this layer from the third-party package molann
(``examples/dipeptide/main.ipynb:31-32,333-348``) and treats it as an opaque
``pp_layer: torch.nn.Module`` (``core.py:65,122,403,414,635``).  This module provides the same
constructor surface (``Feature``, ``FeatureLayer``, ``AlignmentLayer``, ``PreprocessingANN``;
atom groups are duck-typed: ``.ix`` and ``.positions``) on top of :class:`AlignFeatureLayer`,
whose ``forward`` is the hand-written gfx950 kernel K1 behind ``cvf_align_feature_fwd``.

Conventions: unweighted centroid of the align atoms,
``x_aligned = (x - c) @ R`` with ``R = U diag(1,1,sign det(U V^T)) V^T`` from the SVD of
``(x_align - c)^T @ ref_c``; features are concatenated in list order: ``position`` -> 3 n
coordinates (atom-major), ``bond`` -> distance, ``angle`` -> cos (or value), ``dihedral`` ->
(cos, sin) (or value).
"""

import numpy as np
import torch

from . import _hip

_TYPE_ID = {"angle": _hip.FEAT_ANGLE, "bond": _hip.FEAT_BOND, "dihedral": _hip.FEAT_DIHEDRAL,
            "position": _hip.FEAT_POSITION}
_TYPE_NATOMS = {"angle": 3, "bond": 2, "dihedral": 4}


def _ix(atom_group):
    return np.asarray(atom_group.ix if hasattr(atom_group, "ix") else atom_group, dtype=np.int64)


def _batch_records(recs, width=64):
    """Slot records ``[type, s0, s1, s2, s3, out]``, sorted by type, in batches of ``width`` of ONE type (the streaming
    alignment kernel hands a batch to the lanes of a wave: one code path per wave); short batches are padded with
    ``type = -1`` entries (CVF_PP_SLOT_BATCHED)."""
    out, i = [], 0
    while i < len(recs):
        j = i
        while j < len(recs) and recs[j][0] == recs[i][0] and j - i < width:
            j += 1
        out += [list(r) for r in recs[i:j]] + [[-1, 0, 0, 0, 0, 0]] * (width - (j - i))
        i = j
    return out


class Feature:
    """``Feature(name, feature_type, atom_group)`` as in molann (main.ipynb:335)."""

    def __init__(self, name, feature_type, atom_group):
        if feature_type not in _TYPE_ID:
            raise ValueError(f"unknown feature type '{feature_type}' (expected one of {sorted(_TYPE_ID)})")
        idx = _ix(atom_group)
        if feature_type != "position" and len(idx) != _TYPE_NATOMS[feature_type]:
            raise ValueError(f"feature '{name}' of type {feature_type} needs {_TYPE_NATOMS[feature_type]} atoms, got {len(idx)}")
        self.name, self.type_name, self.type_id, self.atom_indices = name, feature_type, _TYPE_ID[feature_type], idx

    def get_name(self):
        return self.name

    def get_type(self):
        return self.type_name

    def get_atom_indices(self):
        return self.atom_indices


class AlignFeatureLayer(torch.nn.Module):
    """Alignment + features on local (0-based, into the input atoms) indices.

    Args:
        n_atoms: number of input atoms N; ``forward`` takes ``[B, N, 3]``.
        align_idx: local indices of the atoms used for the alignment.
        ref_pos: ``[n_align, 3]`` reference positions of those atoms (stored centred).
        features: list of ``(type_name, local_atom_indices)``.
        use_angle_value: emit angles in radians instead of cos / (cos, sin).
        align_weights: optional ``[n_align]`` non-negative per-atom weights w_b of the alignment (masses, say): the
            frame's weighted centroid goes to the reference's and the rotation minimises
            ``sum_b w_b |(x_b - c) R - ref_b|^2``.  ``None`` = the uniform weights of molann's layer.  Weighted layers
            run on the general kernels only (at most 64 atoms per frame).
    """

    def __init__(self, n_atoms, align_idx, ref_pos, features, use_angle_value=False, align_weights=None):
        super().__init__()
        align_idx = np.asarray(align_idx, dtype=np.int64)
        ref = np.asarray(ref_pos, dtype=np.float64)
        assert ref.shape == (len(align_idx), 3), f"ref_pos must be [{len(align_idx)},3], got {ref.shape}"
        assert len(align_idx) >= 3, "at least 3 atoms are needed for the alignment"
        assert align_idx.min() >= 0 and align_idx.max() < n_atoms, "align index out of range"
        rec, out = [], 0
        for tname, atoms in features:
            atoms = [int(a) for a in atoms]
            assert all(0 <= a < n_atoms for a in atoms), f"feature atom index out of range: {atoms}"
            if tname == "position":
                for a in atoms:
                    rec.append([_hip.FEAT_POSITION, a, 0, 0, 0, out])
                    out += 3
            else:
                assert len(atoms) == _TYPE_NATOMS[tname]
                rec.append([_TYPE_ID[tname]] + atoms + [0] * (4 - len(atoms)) + [out])
                out += 2 if (tname == "dihedral" and not use_angle_value) else 1
        self.n_atoms, self.d_r, self.use_angle_value = int(n_atoms), out, bool(use_angle_value)
        self.features = [(t, tuple(int(a) for a in atoms)) for t, atoms in features]
        # structure hints for the kernels (include/cvf.h): affine tables allow the fast kernels
        flags = 0
        if np.array_equal(align_idx, np.arange(len(align_idx))):
            flags |= _hip.PP_ALIGN_CONTIG
        if all(r[0] == _hip.FEAT_POSITION and r[1] == i and r[5] == 3 * i for i, r in enumerate(rec)):
            flags |= _hip.PP_PURE_POSITION
        # per-atom weights: the kernels take them with mean 1, and the reference positions centred by their WEIGHTED
        # centroid and pre-multiplied by them (include/cvf.h, cvf_pp_desc.align_w) - the covariance
        # sum_b w_b (x_b - c) (x) ref_b is then the same loop as the unweighted one
        w_hat = None
        if align_weights is not None:
            w = np.asarray(align_weights, dtype=np.float64).reshape(-1)
            assert w.shape == (len(align_idx),), f"align_weights must be [{len(align_idx)}], got {w.shape}"
            assert np.all(np.isfinite(w)) and np.all(w >= 0) and np.count_nonzero(w) >= 3, \
                "align_weights must be finite, non-negative, with at least 3 non-zero entries"
            assert 3 * n_atoms <= 192, "weighted alignment is built for frames of at most 64 atoms"
            w_hat = w * (len(w) / w.sum())
            flags = 0   # the fast layouts assume uniform weights
        self._flags = flags
        # per-atom tables for the streaming kernel of large molecules (include/cvf.h): which ref row an atom aligns
        # to, and the compact "slot" of every atom some feature reads
        atom_align = np.full(n_atoms, -1, dtype=np.int32)
        atom_align[align_idx] = np.arange(len(align_idx), dtype=np.int32)
        used = sorted({int(a) for r in rec for a in (r[1:2] if r[0] == _hip.FEAT_POSITION else
                                                      r[1:1 + _TYPE_NATOMS[{v: k_ for k_, v in _TYPE_ID.items()}[r[0]]]])})
        atom_slot = np.full(n_atoms, -1, dtype=np.int32)
        atom_slot[used] = np.arange(len(used), dtype=np.int32)
        rec_slot = []
        for r in rec:
            na = 1 if r[0] == _hip.FEAT_POSITION else _TYPE_NATOMS[{v: k_ for k_, v in _TYPE_ID.items()}[r[0]]]
            rec_slot.append([r[0]] + [int(atom_slot[a]) for a in r[1:1 + na]] + [0] * (4 - na) + [r[5]])
        # the streaming kernel hands one record to each lane, 64 at a time: records of one type side by side keep a
        # wave's lanes on one code path (CVF_PP_SLOT_BATCHED; the output offset travels with the record, so the order is
        # free; type -1 entries are padding)
        by_type = sorted(rec_slot, key=lambda r: r[0])
        rec_slot = _batch_records(by_type)
        if w_hat is None:
            self._flags |= _hip.PP_SLOT_BATCHED
        self._n_rec_slot = len(rec_slot)
        self._n_slot = len(used)
        # the derivative kernel of large molecules (csrc/metric_large.hip) scatters J^T g through a table of rows: one row per
        # (record, atom position) pair, the rows of one slot contiguous (include/cvf.h: mrec, slot_row)
        natoms = {_hip.FEAT_POSITION: 1, _hip.FEAT_BOND: 2, _hip.FEAT_ANGLE: 3, _hip.FEAT_DIHEDRAL: 4}
        pairs = sorted((r[1 + j], i, j) for i, r in enumerate(by_type) for j in range(natoms[r[0]]))   # (slot, record, position)
        row_of = {(i, j): n for n, (_, i, j) in enumerate(pairs)}
        slot_row = np.zeros(len(used) + 1, dtype=np.int32)
        for t, _, _ in pairs:
            slot_row[t + 1] += 1
        slot_row = np.cumsum(slot_row).astype(np.int32)
        mrec = []
        for i, r in enumerate(by_type):
            na = natoms[r[0]]
            rows = [row_of[(i, j)] for j in range(na)] + [0] * (4 - na)
            ur = [int(slot_row[r[1 + j]]) for j in range(na)] + [0] * (4 - na)
            sl = list(r[1:5])
            off = [rows[j] - ur[j] if j < na else 0 for j in range(4)]     # position of this record among its atom's rows
            self._row_off_max = max(getattr(self, "_row_off_max", 0), max(off))
            mrec.append([(r[0] + 1) | (r[5] << 3), sl[0] | (sl[1] << 16), sl[2] | (sl[3] << 16), ur[0] | (ur[1] << 16),
                         ur[2] | (ur[3] << 16), off[0] | (off[1] << 8) | (off[2] << 16) | (off[3] << 24), 0, 0])
        self._n_ref = len(pairs)
        self.register_buffer("mrec", torch.tensor(np.asarray(mrec, dtype=np.int64).astype(np.uint32).view(np.int32)
                                                  if mrec else np.zeros((0, 8), np.int32)).reshape(-1, 8))
        self.register_buffer("slot_row", torch.tensor(slot_row, dtype=torch.int32))
        self.register_buffer("atom_align", torch.tensor(atom_align))
        self.register_buffer("atom_slot", torch.tensor(atom_slot))
        self.register_buffer("rec_slot", torch.tensor(rec_slot, dtype=torch.int32).reshape(-1, 6))
        self.register_buffer("slot_atom", torch.tensor(used, dtype=torch.int32))
        self.register_buffer("align_idx", torch.tensor(align_idx, dtype=torch.int32))
        if w_hat is None:
            ref_c = ref - ref.mean(axis=0, keepdims=True)
            self.align_w = None
        else:
            ref_c = w_hat[:, None] * (ref - (w_hat[:, None] * ref).mean(axis=0, keepdims=True))
            self.register_buffer("align_w", torch.tensor(w_hat, dtype=torch.float32))
        self.register_buffer("ref_c", torch.tensor(ref_c, dtype=torch.float32))
        self.register_buffer("rec", torch.tensor(rec, dtype=torch.int32).reshape(-1, 6))

    def derivative_table_limits(self):
        """None, or why the large-molecule derivative kernel (csrc/metric_large.hip: metric_rows_kernel) cannot take this feature
        list: its tables pack a record's slots and rows into 16-bit fields, a record's position among its atom's rows into 8 bits
        and a slot's row count into 12 bits.  EigenFunctionTask's generator mode asks at construction."""
        rows_per_slot = int(np.diff(np.concatenate([[0], self.slot_row.cpu().numpy()])).max()) if self.slot_row.numel() else 0
        for what, have, limit in (("contribution rows (atoms summed over the features)", self._n_ref, 65536),
                                  ("distinct feature atoms", self._n_slot, 65536),
                                  ("features that share one atom", getattr(self, "_row_off_max", 0) + 1, 256),
                                  ("contribution rows of one atom", rows_per_slot, 4096)):
            if have >= limit:
                return f"{have} {what}; the derivative kernel's tables hold fewer than {limit}"
        return None

    def pp_desc(self):
        assert self.rec.is_cuda, "preprocessing layer is not on the GPU (call .to(device))"
        d = _hip.PPDesc()
        d.mode, d.n_coord, d.n_align, d.n_rec = _hip.PP_ALIGN, 3 * self.n_atoms, self.align_idx.numel(), self.rec.shape[0]
        d.d_r, d.use_angle_value = self.d_r, int(self.use_angle_value)
        d.has_position = int(bool((self.rec[:, 0] == _hip.FEAT_POSITION).any().item()))
        d.flags = self._flags
        d.align_idx, d.ref_c, d.rec = self.align_idx.data_ptr(), self.ref_c.data_ptr(), self.rec.data_ptr()
        d.atom_align, d.atom_slot, d.rec_slot = self.atom_align.data_ptr(), self.atom_slot.data_ptr(), self.rec_slot.data_ptr()
        d.slot_atom, d.n_slot, d.n_rec_slot = self.slot_atom.data_ptr(), self._n_slot, self._n_rec_slot
        d.align_w = self.align_w.data_ptr() if self.align_w is not None else None
        if self._n_ref < 65536 and self._n_slot < 65536 and getattr(self, "_row_off_max", 0) < 256:   # (else: cvf_metric_apply names what is missing)
            d.mrec, d.slot_row, d.n_mrec, d.n_ref = self.mrec.data_ptr(), self.slot_row.data_ptr(), self.mrec.shape[0], self._n_ref
        return d

    def forward(self, x):
        """``[B, N, 3]`` -> ``[B, d_r]``.  The reference calls this layer (through ``colvar_model()``) on CPU tensors
        made from the trajectory (2d.ipynb:437-446, main.ipynb:561-562): the frames are moved to the layer's GPU, run
        through kernel K1 and the features come back on the input's device in the input's floating-point type.  No
        autograd: the training tasks use the analytic derivative kernels instead of differentiating through this call."""
        x = torch.as_tensor(x)
        dev = _hip.require_gpu(self.rec.device)   # raises when the layer was never moved to a GPU: there is no CPU path
        assert x.dim() == 3 and x.shape[1] == self.n_atoms and x.shape[2] == 3, \
            f"expected [B,{self.n_atoms},3], got {tuple(x.shape)}"
        if x.requires_grad and torch.is_grad_enabled():
            raise RuntimeError("AlignFeatureLayer.forward is not differentiable through autograd; "
                               "EigenFunctionTask applies its analytic Jacobian on the GPU instead (detach the input)")
        src_dev, src_dt = x.device, (x.dtype if x.dtype.is_floating_point else torch.float32)
        x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        B = x.shape[0]
        out = torch.empty(B, self.d_r, device=dev, dtype=torch.float32)
        if B > 0:
            with torch.cuda.device(dev):
                desc = self.pp_desc()
                scratch = _hip.align_scratch(desc, B, dev)
                _hip.check(_hip.lib().cvf_align_feature_fwd(desc, _hip.ptr(x), B, None, _hip.ptr(out), None, _hip.ptr(scratch),
                                                            _hip.stream()), "cvf_align_feature_fwd")
        return out.to(device=src_dev, dtype=src_dt)


class FeatureLayer:
    """``FeatureLayer(feature_list, input_atom_group, use_angle_value=False)`` (main.ipynb:337)."""

    def __init__(self, feature_list, input_atom_group, use_angle_value=False):
        self.feature_list = list(feature_list)
        self.input_ix = _ix(input_atom_group)
        self.use_angle_value = use_angle_value
        lut = {int(g): i for i, g in enumerate(self.input_ix)}
        self.local = []
        for f in self.feature_list:
            try:
                self.local.append((f.type_name, tuple(lut[int(g)] for g in f.atom_indices)))
            except KeyError as e:
                raise ValueError(f"feature '{f.name}' uses atom {e} that is not in the input atom group") from None

    def get_feature_info(self):
        import pandas as pd
        # molann prints 1-based atom ids in this table (main.ipynb:306-307)
        return pd.DataFrame({"name": [f.name for f in self.feature_list], "type": [f.type_name for f in self.feature_list],
                             "type_id": [f.type_id for f in self.feature_list],
                             "atom indices": [list(f.atom_indices + 1) for f in self.feature_list]})

    def output_dimension(self):
        return sum(3 * len(a) if t == "position" else (2 if t == "dihedral" and not self.use_angle_value else 1)
                   for t, a in self.local)


class AlignmentLayer:
    """``AlignmentLayer(align_atom_group, input_atom_group)`` (main.ipynb:345)."""

    def __init__(self, align_atom_group, input_atom_group, weights=None):
        self.weights = None if weights is None else np.asarray(weights, dtype=np.float64)
        self.align_ix = _ix(align_atom_group)
        self.input_ix = _ix(input_atom_group)
        lut = {int(g): i for i, g in enumerate(self.input_ix)}
        try:
            self.local_idx = np.asarray([lut[int(g)] for g in self.align_ix], dtype=np.int64)
        except KeyError as e:
            raise ValueError(f"align atom {e} is not in the input atom group") from None
        pos = np.asarray(align_atom_group.positions, dtype=np.float64)
        self.ref_c = pos - pos.mean(axis=0, keepdims=True)   # (the fused layer re-centres by the weighted centroid)

    def show_info(self):
        print(f"\n{len(self.input_ix)} atoms used for input, (0-based) global indices: \n {list(self.input_ix)}")
        print(f"\n{len(self.align_ix)} atoms used for alignment, with (0-based) global indices: \n {list(self.align_ix)}")
        print("local indices\n", list(self.local_idx))
        print("\npositions of reference state used in aligment:\n", self.ref_c.astype(np.float32))


class PreprocessingANN(AlignFeatureLayer):
    """``PreprocessingANN(align_layer, feature_layer)`` (main.ipynb:348) -> one fused GPU layer."""

    def __init__(self, align_layer, feature_layer):
        assert np.array_equal(align_layer.input_ix, feature_layer.input_ix), "align and feature layers use different input atoms"
        super().__init__(len(align_layer.input_ix), align_layer.local_idx, align_layer.ref_c, feature_layer.local,
                         feature_layer.use_angle_value, align_weights=getattr(align_layer, "weights", None))


def identity_desc(n_coord):
    d = _hip.PPDesc()
    d.mode, d.n_coord, d.d_r = _hip.PP_IDENTITY, int(n_coord), int(n_coord)
    return d
