"""Deterministic synthetic tiles, alphas and initial weights (numpy PCG64 only).

The hot path is benchmarked and parity-tested on synthetic data of the reference's
shapes: the reference z-scores every field
(reference DoWnGAN/helpers/gen_experiment_datasets.py:195-201), so i.i.d. N(0,1)
tiles are distribution-faithful.  Weights follow torch's default Conv2d/Linear
initialisation (kaiming-uniform with a=sqrt(5) == U(+-1/sqrt(fan_in)); bias
U(+-1/sqrt(fan_in))), but are drawn from numpy so fixtures do not depend on the
torch RNG stream or version.

Parameter names and shapes are the reference's ``state_dict`` contract
(reference DoWnGAN/networks/generator.py:58-81, DoWnGAN/networks/critic.py:12-99).
"""
from __future__ import annotations

import numpy as np

WEIGHT_SEED = 0
DATA_SEED = 1234
ALPHA_SEED = 4321


def generator_param_specs(filters, channels, n_predictands=2, num_res_blocks=16, num_upsample=3):
    """Ordered [(name, shape)] of the reference Generator (generator.py:58-81)."""
    specs = []

    def conv(name, cout, cin):
        specs.append((name + ".weight", (cout, cin, 3, 3)))
        specs.append((name + ".bias", (cout,)))

    conv("conv1", filters, channels)
    for i in range(num_res_blocks):
        for j in range(3):
            for k in range(1, 6):
                conv(f"res_blocks.{i}.dense_blocks.{j}.b{k}.0", filters, k * filters)
    conv("conv2", filters, filters)
    for u in range(num_upsample):
        conv(f"upsampling.{3 * u}", 4 * filters, filters)
    conv("conv3.0", filters, filters)
    conv("conv3.2", n_predictands, filters)
    return specs


def critic_param_specs(coarse_dim, fine_dim, nc):
    """Ordered [(name, shape)] of the reference Critic (critic.py:20-99)."""
    cd = coarse_dim
    widths = [(cd, nc), (cd, cd), (2 * cd, cd), (2 * cd, 2 * cd), (4 * cd, 2 * cd),
              (4 * cd, 4 * cd), (8 * cd, 4 * cd), (8 * cd, 8 * cd)]
    specs = []
    for li, (co, ci) in enumerate(widths):
        specs.append((f"features.{2 * li}.weight", (co, ci, 3, 3)))
        if li == 0:
            specs.append(("features.0.bias", (co,)))
    fc_in = int((cd * 2 ** 3) * (fine_dim / 2 ** 4) ** 2)  # critic.py:95
    specs.append(("classifier.0.weight", (100, fc_in)))
    specs.append(("classifier.0.bias", (100,)))
    specs.append(("classifier.2.weight", (1, 100)))
    specs.append(("classifier.2.bias", (1,)))
    return specs


def _fan_in(name, shape, prev_weight_shape):
    if name.endswith(".weight"):
        return int(np.prod(shape[1:]))
    return int(np.prod(prev_weight_shape[1:]))


def init_params(specs, seed=WEIGHT_SEED, stream=0):
    """{name: float32 ndarray} drawn U(+-1/sqrt(fan_in)), one PCG64 stream per tensor."""
    out = {}
    prev_w = None
    for idx, (name, shape) in enumerate(specs):
        if name.endswith(".weight"):
            prev_w = shape
        bound = 1.0 / np.sqrt(_fan_in(name, shape, prev_w))
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, stream, idx])))
        a = rng.random(size=shape, dtype=np.float32)
        out[name] = ((a * 2.0 - 1.0) * np.float32(bound)).astype(np.float32)
    return out


def generator_params(filters, channels, n_predictands=2, num_res_blocks=16, num_upsample=3, seed=WEIGHT_SEED):
    return init_params(generator_param_specs(filters, channels, n_predictands, num_res_blocks, num_upsample),
                       seed, stream=1)


def critic_params(coarse_dim, fine_dim, nc, seed=WEIGHT_SEED):
    return init_params(critic_param_specs(coarse_dim, fine_dim, nc), seed, stream=2)


def tiles(batch, channels, coarse_side, n_predictands=2, seed=DATA_SEED, rank=0, mask_channel=None):
    """(coarse [B,C,S,S], fine [B,P,8S,8S]) float32 i.i.d. N(0,1).

    ``mask_channel`` optionally makes one covariate a Bernoulli(0.5) {0,1} field, like the
    un-normalised land-sea mask slot (reference gen_experiment_datasets.py:209).
    """
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed + rank, 11])))
    coarse = rng.standard_normal((batch, channels, coarse_side, coarse_side), dtype=np.float32)
    fine = rng.standard_normal((batch, n_predictands, 8 * coarse_side, 8 * coarse_side), dtype=np.float32)
    if mask_channel is not None:
        coarse[:, mask_channel] = (rng.random((batch, coarse_side, coarse_side)) < 0.5).astype(np.float32)
    return coarse, fine


def alpha(batch, step, seed=ALPHA_SEED, rank=0):
    """alpha[B] ~ U[0,1): the host-injected replacement of torch.rand in _gp (wasserstein.py:91)."""
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed + step, 13, rank])))
    return rng.random(batch, dtype=np.float32)
