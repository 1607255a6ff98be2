"""Oracle restatement of ``colvarsfinder/nn.py`` as functions over state dicts.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Networks are dictionaries
``name -> tensor`` with the reference's ``state_dict`` key layout:

* ``create_sequential_nn`` (nn.py:29-59): ``Linear`` modules named '1'..'L', the
  activation after every layer but the last -> keys ``'<l>.weight' [out,in]``,
  ``'<l>.bias' [out]``.
* ``EigenFunctions`` (nn.py:242-293): k such networks under ``eigen_funcs.<i>.``,
  outputs concatenated along dim 1 (nn.py:293).
* ``AutoEncoder`` (nn.py:61-114): ``encoder.`` / ``decoder.`` prefixes,
  ``forward = decoder(encoder(x))`` (nn.py:114).
"""

import math
import torch


def init_sequential(layer_dims, prefix, generator=None, dtype=torch.float32):
    """torch.nn.Linear's default init (kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(in), 1/sqrt(in)))."""
    sd = {}
    for l in range(len(layer_dims) - 1):
        fan_in, fan_out = layer_dims[l], layer_dims[l + 1]
        bound = 1.0 / math.sqrt(fan_in)
        w = (torch.rand(fan_out, fan_in, generator=generator, dtype=torch.float32) * 2 - 1) * bound
        b = (torch.rand(fan_out, generator=generator, dtype=torch.float32) * 2 - 1) * bound
        sd[f"{prefix}{l + 1}.weight"] = w.to(dtype)
        sd[f"{prefix}{l + 1}.bias"] = b.to(dtype)
    return sd


def init_eigenfunctions(layer_dims, k, generator=None, dtype=torch.float32):
    sd = {}
    for i in range(k):
        sd.update(init_sequential(layer_dims, f"eigen_funcs.{i}.", generator, dtype))
    return sd


def init_autoencoder(e_dims, d_dims, generator=None, dtype=torch.float32):
    assert e_dims[-1] == d_dims[0]
    sd = init_sequential(e_dims, "encoder.", generator, dtype)
    sd.update(init_sequential(d_dims, "decoder.", generator, dtype))
    return sd


def init_regautoencoder(e_dims, d_dims, r_dims, K, generator=None, dtype=torch.float32):
    """``RegAutoEncoder`` (nn.py:116-198): ``encoder.`` / ``decoder.`` / ``reg.<i>.`` prefixes, in module order."""
    assert e_dims[-1] == d_dims[0] and (K == 0 or e_dims[-1] == r_dims[0])
    sd = init_sequential(e_dims, "encoder.", generator, dtype)
    sd.update(init_sequential(d_dims, "decoder.", generator, dtype))
    for i in range(K):
        sd.update(init_sequential(r_dims, f"reg.{i}.", generator, dtype))
    return sd


def n_layers(sd, prefix):
    l = 0
    while f"{prefix}{l + 1}.weight" in sd:
        l += 1
    return l


def sequential_forward(sd, prefix, x, activation=torch.tanh):
    """nn.py:52-59: activation after every Linear except the last."""
    L = n_layers(sd, prefix)
    h = x
    for l in range(1, L + 1):
        h = torch.nn.functional.linear(h, sd[f"{prefix}{l}.weight"], sd[f"{prefix}{l}.bias"])
        if l < L:
            h = activation(h)
    return h


def eigenfunctions_forward(sd, k, x, activation=torch.tanh):
    """nn.py:293."""
    return torch.cat([sequential_forward(sd, f"eigen_funcs.{i}.", x, activation) for i in range(k)], dim=1)


def autoencoder_forward(sd, x, activation=torch.tanh):
    """nn.py:114."""
    return sequential_forward(sd, "decoder.", sequential_forward(sd, "encoder.", x, activation), activation)


def encoder_forward(sd, x, activation=torch.tanh):
    return sequential_forward(sd, "encoder.", x, activation)


def reorder_eigenfunctions(sd, cvec):
    """core.py:356-370: new dict whose net j is the old net cvec[j]."""
    out = {}
    for j, src in enumerate(int(c) for c in cvec):
        pre = f"eigen_funcs.{src}."
        for key, val in sd.items():
            if key.startswith(pre):
                out[f"eigen_funcs.{j}." + key[len(pre):]] = val.clone()
    return out


def regautoencoder_forward_ae(sd, x, activation=torch.tanh):
    """``RegAutoEncoder.forward_ae`` (nn.py:164-172)."""
    return sequential_forward(sd, "decoder.", sequential_forward(sd, "encoder.", x, activation), activation)


def regautoencoder_forward_reg(sd, K, x, activation=torch.tanh):
    """``RegAutoEncoder.forward_reg`` (nn.py:174-186)."""
    z = sequential_forward(sd, "encoder.", x, activation)
    return torch.cat([sequential_forward(sd, f"reg.{i}.", z, activation) for i in range(K)], dim=1)
