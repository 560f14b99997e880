"""Divergence / vorticity metrics: the oracle restatement is pinned by the known answers of the reference's own tests."""
import numpy as np
import torch

from oracle import physics


def test_oracle_reproduces_reference_known_answers():
    hr, fake = physics.reference_test_fixture()
    assert hr.shape == (64, 2, 10, 12)
    assert np.isclose(physics.divergence_loss(hr, fake), physics.KNOWN["divergence"], atol=physics.KNOWN["atol"])
    assert np.isclose(physics.vorticity_loss(hr, fake), physics.KNOWN["vorticity"], atol=physics.KNOWN["atol"])


def test_identical_fields_have_zero_loss_and_scale_invariance():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 2, 17, 19, generator=g)
    y = torch.randn(3, 2, 17, 19, generator=g)
    assert physics.divergence_loss(x, x) == 0.0 and physics.vorticity_loss(x, x) == 0.0
    a, b = physics.divergence_loss(x, y), physics.divergence_loss(3.0 * x, 0.5 * y)      # each side is divided by its own std
    assert abs(a - b) < 1e-5 * a
