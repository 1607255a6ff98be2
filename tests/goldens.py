"""Helpers to read the fixtures under tests/golden/ (written by tools/gen_golden.py)."""

import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

KAT_CASES = ["kat_gen_id2_k1", "kat_gen_id2_k3", "kat_tr_id2_k2", "kat_gen_mol10_k2", "kat_gen_mol22_k3",
             "kat_tr_mol10_k2", "kat_gen_mix10_k2"]
EF_TRAIN_CASES = ["train_gen_id2_k1", "train_tr_id2_k2", "train_gen_mol22_k3", "train_tr_mol10_k2"]
AE_TRAIN_CASES = ["train_ae_id2", "train_ae_mol22"]
REGAE_TRAIN_CASES = ["train_regae_id2_k1", "train_regae_mol10_k2", "train_regae_id2_k2_frozen", "train_regae_mol10_k2_eta",
                     "train_regae_id2_k1_eta", "train_regae_mol10_k2_eg", "train_regae_id2_k1_eg", "train_regae_id2_k1_gen",
                     "train_regae_mol10_k2_gen", "train_regae_id2_k2_gen_frozen"]


def load(name, tag):
    return np.load(os.path.join(GOLDEN, f"{name}_{tag}.npz"), allow_pickle=False)


def state_dict(g, prefix="sd/", dtype=None):
    out = {}
    for key in g.files:
        if key.startswith(prefix):
            t = torch.from_numpy(np.array(g[key]))
            out[key[len(prefix):]] = t.to(dtype) if dtype is not None else t
    return out


def features_list(g):
    """Rebuild the [(type, atoms)] list stored by gen_golden.pp_meta."""
    types = [str(t) for t in g["feat_types"]]
    atoms = g["feat_atoms"]
    pos_atoms = list(int(i) for i in g["feat_pos_atoms"])
    feats, p = [], 0
    n_pos = sum(1 for t in types if t == "position")
    for t, row in zip(types, atoms):
        if t == "position":
            assert n_pos == 1, "fixtures hold at most one position feature"
            feats.append((t, tuple(pos_atoms)))
        else:
            feats.append((t, tuple(int(i) for i in row if i >= 0)))
    return feats


def pp_spec(g):
    """dict describing the preprocessing layer of a fixture (None for identity)."""
    if str(g["pp"]) == "identity":
        return None
    return dict(align_idx=[int(i) for i in g["align_idx"]], ref_pos=np.array(g["ref_pos"]),
                features=features_list(g), use_angle_value=bool(g["use_angle_value"]))


# Bench-sized fixtures (tools/gen_golden.py run_big_cases: BASELINE configs 2 / 3 at 100 000 frames, batches of 20 000): the
# file holds the arguments of tests.synth.make_molecule_traj instead of the frames ("seed + outputs only").
BIG_EF_CASES = ["big_gen_c3", "big_tr_c3"]
BIG_AE_CASES = ["big_ae_c2"]


class Synth:
    """An npz fixture whose ``traj`` / ``w`` entries are regenerated from ``traj_gen`` = (atoms, frames, seed)."""

    def __init__(self, g):
        from tests.synth import make_molecule_traj
        self._g = g
        n_atoms, n_frames, seed = (int(v) for v in g["traj_gen"])
        traj, w, ref = make_molecule_traj(n_atoms, n_frames, seed)
        assert np.array_equal(ref, np.array(g["ref_pos"])), "make_molecule_traj no longer reproduces the fixture's reference structure"
        self._extra = {"traj": traj, "w": w}
        self.files = list(g.files) + ["traj", "w"]

    def __getitem__(self, key):
        return self._extra[key] if key in self._extra else self._g[key]


def load_big(name, tag):
    return Synth(load(name, tag))
