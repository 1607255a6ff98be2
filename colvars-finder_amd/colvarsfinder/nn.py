"""Network containers with the module/parameter layout of the reference's ``colvarsfinder.nn``.

API twin of reference ``colvarsfinder/nn.py`` (own code): same class names, constructor
arguments, attribute names and ``state_dict`` keys, so models, checkpoints and user scripts
are interchangeable:

* ``create_sequential_nn`` (nn.py:29-59): ``Linear`` children named ``'1'..'L'``, the
  activation (ONE shared instance) registered as ``'activation i'`` after every layer but
  the last.
* ``EigenFunctions`` (nn.py:242-293), ``AutoEncoder`` (nn.py:61-114),
  ``RegAutoEncoder`` (nn.py:116-203), ``RegModel`` (nn.py:205-239).

These classes are parameter containers plus a plain ``forward`` (used for inference and
TorchScript export).  Training on the MI355X never calls ``forward``: the tasks in
``colvarsfinder.core`` flatten the parameters into one HBM buffer and run the hand-written
HIP kernels on it (``mlp_layout`` below describes that buffer to the C ABI).
"""

import re

import numpy as np
import torch


def create_sequential_nn(layer_dims, activation=torch.nn.Tanh()):
    """Feed-forward net ``layer_dims[0] -> ... -> layer_dims[-1]`` (activation between layers only)."""
    n = len(layer_dims)
    assert n >= 2, f"Error: at least 2 layers are needed to define a neural network (length={n})!"
    net = torch.nn.Sequential()
    for pos, (d_in, d_out) in enumerate(zip(layer_dims[:-1], layer_dims[1:]), start=1):
        net.add_module(str(pos), torch.nn.Linear(d_in, d_out))
        if pos < n - 1:
            net.add_module(f"activation {pos}", activation)
    return net


def _encoder_cv_params(encoder, n_encoder_layers, cv_idx):
    """Named parameters of an encoder restricted to output ``cv_idx`` (last layer sliced to one row)."""
    picked = []
    for name, param in encoder.named_parameters():
        layer = int(re.search(r"\d+", name).group())
        picked.append([name, param if layer < n_encoder_layers else param[cv_idx:cv_idx + 1, ...]])
    return picked


class AutoEncoder(torch.nn.Module):
    """``encoder`` / ``decoder`` pair; ``forward(x) = decoder(encoder(x))``."""

    def __init__(self, e_layer_dims, d_layer_dims, activation=torch.nn.Tanh()):
        super().__init__()
        assert e_layer_dims[-1] == d_layer_dims[0], "ouput dimension of encoder and input dimension of decoder do not match!"
        self.encoder = create_sequential_nn(e_layer_dims, activation)
        self.decoder = create_sequential_nn(d_layer_dims, activation)
        self.encoded_dim = e_layer_dims[-1]
        self._num_encoder_layer = len(e_layer_dims) - 1

    def get_params_of_cv(self, cv_idx):
        assert 0 <= cv_idx < self.encoded_dim, f"index {cv_idx} exceeded the range [0, {self.encoded_dim-1}]!"
        return _encoder_cv_params(self.encoder, self._num_encoder_layer, cv_idx)

    def forward(self, inp):
        return self.decoder(self.encoder(inp))


class RegAutoEncoder(torch.nn.Module):
    """Autoencoder plus ``K`` scalar regulariser nets on the latent space."""

    def __init__(self, e_layer_dims, d_layer_dims, reg_layer_dims, K, activation=torch.nn.Tanh()):
        super().__init__()
        assert e_layer_dims[-1] == d_layer_dims[0], "ouput dimension of encoder and input dimension of decoder do not match!"
        self.num_reg = K
        assert K == 0 or e_layer_dims[-1] == reg_layer_dims[0], \
            "ouput dimension of encoder and input dimension of regulator part do not match!"
        self.encoder = create_sequential_nn(e_layer_dims, activation)
        self.decoder = create_sequential_nn(d_layer_dims, activation)
        self.encoded_dim = e_layer_dims[-1]
        self._num_encoder_layer = len(e_layer_dims) - 1
        self.reg = torch.nn.ModuleList(create_sequential_nn(reg_layer_dims, activation) for _ in range(K)) if K > 0 else None

    def get_params_of_cv(self, cv_idx):
        assert 0 <= cv_idx < self.encoded_dim, f"index {cv_idx} exceeded the range [0, {self.encoded_dim-1}]!"
        return _encoder_cv_params(self.encoder, self._num_encoder_layer, cv_idx)

    def forward_ae(self, inp):
        return self.decoder(self.encoder(inp))

    def forward_reg(self, inp):
        assert self.num_reg > 0, "number of regularizers is not positive."
        z = self.encoder(inp)
        return torch.cat([net(z) for net in self.reg], dim=1)

    def forward(self, inp):
        z = self.encoder(inp)
        return torch.cat((self.decoder(z), torch.cat([net(z) for net in self.reg], dim=1)), dim=1)


class RegModel(torch.nn.Module):
    """Regulariser nets of a :class:`RegAutoEncoder` evaluated in the order ``cvec``."""

    def __init__(self, reg_ae, cvec):
        super().__init__()
        assert reg_ae.num_reg > 0, "number of regularizers is not positive."
        assert len(cvec) == reg_ae.num_reg, "length of cvec doesn't equal to number of regularizers"
        assert (sorted(cvec) == np.arange(reg_ae.num_reg)).all(), f"cvec should be a permutation of 0,1,...,{len(cvec)-1}."
        self.encoder = reg_ae.encoder
        self.reg = reg_ae.reg
        self.cvec = cvec
        self.encoded_dim = reg_ae.encoded_dim
        self.num_reg = reg_ae.num_reg

    def forward(self, inp):
        z = self.encoder(inp)
        return torch.cat([self.reg[idx](z) for idx in self.cvec], dim=1)


class EigenFunctions(torch.nn.Module):
    """``k`` scalar nets of identical architecture; ``forward`` concatenates their outputs to ``[l, k]``."""

    def __init__(self, layer_dims, k, activation=torch.nn.Tanh()):
        super().__init__()
        assert layer_dims[-1] == 1, "each eigenfunction must be scalar-valued"
        self.eigen_funcs = torch.nn.ModuleList(create_sequential_nn(layer_dims, activation) for _ in range(k))

    def get_params_of_cv(self, cv_idx):
        return [[name, param] for name, param in self.eigen_funcs[cv_idx].named_parameters()]

    def forward(self, inp):
        return torch.cat([net(inp) for net in self.eigen_funcs], dim=1)


# ----------------------------------------------------------------------------------------------
# Flat-buffer description of a model for the HIP kernels (not part of the reference API)
# ----------------------------------------------------------------------------------------------

# activation codes of include/cvf.h (cvf_mlp_desc.act); 1 == True, so "followed by Tanh" still reads as a flag
ACT_NONE, ACT_TANH, ACT_SIGMOID, ACT_RELU, ACT_ELU, ACT_LEAKY_RELU, ACT_SOFTPLUS = range(7)


def _act_code(module):
    """cvf_mlp_desc.act code of a torch activation module (the reference takes any, nn.py:29-59); the chain kernels of the
    autoencoder tasks implement these, with the modules' default parameters."""
    if isinstance(module, torch.nn.Tanh):
        return ACT_TANH
    if isinstance(module, torch.nn.Sigmoid):
        return ACT_SIGMOID
    if isinstance(module, torch.nn.ReLU):
        return ACT_RELU
    if isinstance(module, torch.nn.ELU) and module.alpha == 1.0:
        return ACT_ELU
    if isinstance(module, torch.nn.LeakyReLU) and module.negative_slope == 0.01:
        return ACT_LEAKY_RELU
    if isinstance(module, torch.nn.Softplus) and module.beta == 1.0 and module.threshold == 20.0:
        return ACT_SOFTPLUS
    raise NotImplementedError(
        f"the MI355X kernels implement Tanh, Sigmoid, ReLU, ELU(alpha=1), LeakyReLU(0.01) and Softplus(beta=1, threshold=20) "
        f"(got {module})")


def _chain_layers(seq):
    """[(Linear, activation code that follows it - 0: none)] of a create_sequential_nn module; rejects anything else."""
    out = []
    children = list(seq._modules.values())  # children() would drop the repeats of the shared activation
    for i, child in enumerate(children):
        if isinstance(child, torch.nn.Linear):
            nxt = children[i + 1] if i + 1 < len(children) else None
            has_act = nxt is not None and not isinstance(nxt, torch.nn.Linear)
            out.append((child, _act_code(nxt) if has_act else ACT_NONE))
    return out


def mlp_layout(model):
    """Describe ``model`` as chains of Linear layers over ONE flat fp32 buffer laid out in
    ``model.parameters()`` order.  Returns dict(nets=[[(w_off, b_off, in, out, act), ...], ...], n_params)."""
    offsets, pos = {}, 0
    for p in model.parameters():
        offsets[id(p)] = pos
        pos += p.numel()
    if isinstance(model, EigenFunctions):
        chains = [_chain_layers(net) for net in model.eigen_funcs]
        # (the eigenfunction kernels need the activation's first two derivatives: the 16-frame kernels carry tanh's, the
        #  64-frame kernels take every code above - csrc/ef_mfma.hip: ef_shape; other shapes are rejected by the task)
    elif isinstance(model, AutoEncoder):
        chains = [_chain_layers(model.encoder) + _chain_layers(model.decoder)]
    else:
        raise TypeError(f"no HIP layout for model type {type(model).__name__}")
    nets = [[(offsets[id(lin.weight)], offsets[id(lin.bias)], lin.in_features, lin.out_features, int(act))
             for lin, act in chain] for chain in chains]
    return dict(nets=nets, n_params=pos)
