"""ResidentLoader on the GPU: one gather launch per tensor forms the native minibatch; a train epoch runs from it."""
import pytest
import torch

from downgan_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gather_forms_native_batches(dtype):
    from downgan_amd.GAN.dataloader import NetCDFSR, ResidentLoader, epoch_indices
    g = torch.Generator().manual_seed(0)
    coarse, fine = torch.randn(11, 6, 8, 8, generator=g), torch.randn(11, 2, 64, 64, generator=g)
    ld = ResidentLoader(NetCDFSR(coarse, fine), batch_size=3, dtype=dtype, seed=5)
    order = epoch_indices(11, 3, 1, 0, 5)
    seen = 0
    for (xc, xf), idx in zip(ld, order):
        idx = idx[0]
        assert xc.shape == (3, 6, 8, 8) and xf.shape == (3, 2, 64, 64)
        for nat, ref, c in ((xc.nhwc, coarse, 6), (xf.nhwc, fine, 2)):
            want = ref[idx].permute(0, 2, 3, 1).to(nat.dtype)
            assert torch.equal(nat[..., :c].cpu(), want)                      # bit-exact copy of the staged values
            assert float(nat[..., c:].abs().max()) == 0.0                     # padding channels stay zero
        seen += 1
    assert seen == len(ld) == 3


def test_train_epoch_from_resident_loader_matches_tensor_batches():
    """The same two global batches through (a) ResidentLoader native batches and (b) NCHW tensors give the same losses."""
    from downgan_amd.GAN.dataloader import NetCDFSR, ResidentLoader, epoch_indices
    from downgan_amd.GAN.wasserstein import WassersteinGAN
    from downgan_amd.networks.critic import Critic
    from downgan_amd.networks.generator import Generator
    coarse, fine = synthetic.tiles(8, 2, 16)
    coarse, fine = torch.from_numpy(coarse), torch.from_numpy(fine)
    alphas = [torch.from_numpy(synthetic.alpha(4, s)) for s in range(2)]
    logs = []
    for native in (True, False):
        G, C = Generator(16, 128, 2, 2, num_res_blocks=2, dtype="f32"), Critic(16, 128, 2, dtype="f32")
        tr = WassersteinGAN(G, C)
        if native:
            batches = list_batches = ResidentLoader(NetCDFSR(coarse, fine), batch_size=4, dtype="f32", seed=1)
        else:
            order = epoch_indices(8, 4, 1, 0, 1)
            batches = [(coarse[i[0]], fine[i[0]]) for i in order]
        out = []
        for s, (c, f) in enumerate(batches):
            out.append(dict(tr._critic_train_iteration(c, f, alpha=alphas[s])))
            tr.num_steps += 1
        logs.append(out)
    for a, b in zip(*logs):
        for k in ("critic_loss", "gp_ret", "c_real_mean", "c_fake_mean"):
            assert abs(a[k] - b[k]) <= 1e-5 * max(abs(a[k]), 1e-3), (k, a[k], b[k])
