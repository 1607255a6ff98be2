"""TorchScript export of the collective-variable model (reference ``save_model``, core.py:205-226).

The reference scripts ``colvar_model() = Sequential(pp_layer, nets)`` and saves ``scripted_cv_cpu.pt`` /
``scripted_cv_gpu.pt`` for downstream programs (e.g. MD engines evaluating the learned CV through libtorch).
The training path of this package runs the alignment + feature layer as a gfx950 kernel behind a C ABI, which
TorchScript cannot serialise; :class:`ScriptableAlignFeature` is its *export twin*: the same map written with
torch operators only, so that the saved file is self-contained and runs wherever libtorch runs.

It is an export artefact, not a fallback: nothing in the training or evaluation path of this package
instantiates it (``save_model`` builds it for scripting and discards it).  Conventions as in ``pp.py``:
unweighted centroid of the align atoms, ``x_aligned = (x - c) @ R`` with
``R = U diag(1, 1, sign det(U V^T)) V^T`` from the SVD of ``(x_align - c)^T ref_c``; features in list order.
"""

from typing import List

import torch

from . import _hip


class ScriptableAlignFeature(torch.nn.Module):
    """Pure-torch, TorchScript-compatible alignment + feature map built from an ``AlignFeatureLayer``."""

    def __init__(self, layer):
        super().__init__()
        rec = layer.rec.detach().cpu().to(torch.long)
        self.d_r: int = int(layer.d_r)
        self.use_angle_value: bool = bool(layer.use_angle_value)
        self.register_buffer("align_idx", layer.align_idx.detach().cpu().to(torch.long))
        self.register_buffer("ref_c", layer.ref_c.detach().cpu().clone())
        # per-atom alignment weights (mean 1; ref_c is already multiplied by them, pp.AlignFeatureLayer); ones = uniform
        aw = getattr(layer, "align_w", None)
        self.register_buffer("align_w", torch.ones(layer.ref_c.shape[0]) if aw is None else aw.detach().cpu().clone())

        def pick(type_id, n_atoms):
            rows = rec[rec[:, 0] == type_id]
            return rows[:, 1:1 + n_atoms].contiguous(), rows[:, 5].contiguous()

        pos_a, pos_o = pick(_hip.FEAT_POSITION, 1)
        bond_a, bond_o = pick(_hip.FEAT_BOND, 2)
        ang_a, ang_o = pick(_hip.FEAT_ANGLE, 3)
        dih_a, dih_o = pick(_hip.FEAT_DIHEDRAL, 4)
        self.register_buffer("pos_atoms", pos_a.reshape(-1))
        self.register_buffer("pos_out", (pos_o[:, None] + torch.arange(3)[None, :]).reshape(-1))
        self.register_buffer("bond_atoms", bond_a)
        self.register_buffer("bond_out", bond_o)
        self.register_buffer("ang_atoms", ang_a)
        self.register_buffer("ang_out", ang_o)
        self.register_buffer("dih_atoms", dih_a)
        self.register_buffer("dih_out", dih_o)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ref = self.ref_c.to(x.dtype)
        out = torch.zeros(x.shape[0], self.d_r, dtype=x.dtype, device=x.device)
        if self.pos_atoms.numel() > 0:
            xa = x[:, self.align_idx, :]
            c = (self.align_w.to(x.dtype)[None, :, None] * xa).mean(dim=1, keepdim=True)
            h = torch.matmul((xa - c).transpose(1, 2), ref)
            u, s, vh = torch.linalg.svd(h)
            d = torch.sign(torch.linalg.det(torch.matmul(u, vh)))
            ones = torch.ones_like(d)
            r = torch.matmul(u * torch.stack([ones, ones, d], dim=-1)[:, None, :], vh)
            al = torch.matmul(x[:, self.pos_atoms, :] - c, r)
            out[:, self.pos_out] = al.reshape(x.shape[0], -1)
        if self.bond_out.numel() > 0:
            d01 = x[:, self.bond_atoms[:, 1], :] - x[:, self.bond_atoms[:, 0], :]
            out[:, self.bond_out] = torch.sqrt((d01 * d01).sum(-1))
        if self.ang_out.numel() > 0:
            r1 = x[:, self.ang_atoms[:, 0], :] - x[:, self.ang_atoms[:, 1], :]
            r2 = x[:, self.ang_atoms[:, 2], :] - x[:, self.ang_atoms[:, 1], :]
            cs = (r1 * r2).sum(-1) / torch.sqrt((r1 * r1).sum(-1) * (r2 * r2).sum(-1))
            out[:, self.ang_out] = torch.acos(cs) if self.use_angle_value else cs
        if self.dih_out.numel() > 0:
            b1 = x[:, self.dih_atoms[:, 1], :] - x[:, self.dih_atoms[:, 0], :]
            b2 = x[:, self.dih_atoms[:, 2], :] - x[:, self.dih_atoms[:, 1], :]
            b3 = x[:, self.dih_atoms[:, 3], :] - x[:, self.dih_atoms[:, 2], :]
            n1 = torch.cross(b1, b2, dim=-1)
            n2 = torch.cross(b2, b3, dim=-1)
            inv = 1.0 / torch.sqrt((n1 * n1).sum(-1) * (n2 * n2).sum(-1))
            cs = (n1 * n2).sum(-1) * inv
            sn = (n1 * b3).sum(-1) * torch.sqrt((b2 * b2).sum(-1)) * inv
            if self.use_angle_value:
                out[:, self.dih_out] = torch.atan2(sn, cs)
            else:
                out[:, self.dih_out] = cs
                out[:, self.dih_out + 1] = sn
        return out


def scriptable_cv(cv: torch.nn.Sequential) -> torch.nn.Sequential:
    """``colvar_model()`` with a kernel-backed preprocessing layer replaced by its export twin (deep copies on CPU)."""
    import copy
    mods: List[torch.nn.Module] = []
    for m in cv.children():
        if hasattr(m, "pp_desc") and hasattr(m, "rec"):
            mods.append(ScriptableAlignFeature(m))
        else:
            mods.append(copy.deepcopy(m).to("cpu"))
    return torch.nn.Sequential(*mods)
