"""1-bit LeakyReLU' masks (dg_epilogue.mask_bits / out_bits) on the host-side emulation: the critic with bit masks gives
exactly the results of the critic that re-reads its activations as masks."""
import os

import torch

from downgan_amd import synthetic
from downgan_amd.engine import HyperParams, TrainEngine
from downgan_amd.layout import nchw_to_nhwc_padded
from oracle.emu_ops import EmuOps


def run(bits):
    if bits:
        os.environ.pop("DG_NO_MASK_BITS", None)
    else:
        os.environ["DG_NO_MASK_BITS"] = "1"
    try:
        ops = EmuOps("f32")
        B, S, F_ = 1, 16, 128                     # critic widths 128..1024: the bit path's 64-channel blocks
        eng = TrainEngine(ops, S, F_, 2, B, HyperParams(batch_size=B), num_res_blocks=1)
        assert (eng.C.act_bits is not None) == bits
        eng.G.load_state_dict(synthetic.generator_params(F_, 2, 2, 1))
        eng.C.load_state_dict(synthetic.critic_params(F_, 8 * S, 2))
        coarse, fine = synthetic.tiles(B, 2, S)
        xc = nchw_to_nhwc_padded(torch.from_numpy(coarse), eng.G.cin_p, ops.tdtype)
        xf = nchw_to_nhwc_padded(torch.from_numpy(fine), eng.G.np_p, ops.tdtype)
        eng.critic_iteration(xc, xf, torch.from_numpy(synthetic.alpha(B, 0)), apply_update=False)
        return eng.read_scalars(), eng.C.grad_dict()
    finally:
        os.environ.pop("DG_NO_MASK_BITS", None)


def test_bit_masks_reproduce_activation_masks_exactly():
    torch.set_num_threads(8)
    s1, g1 = run(True)
    s0, g0 = run(False)
    for k in ("c_real_mean", "c_fake_mean", "gp_ret", "critic_loss"):
        assert s1[k] == s0[k], (k, s1[k], s0[k])
    for k in g0:
        assert torch.equal(g1[k], g0[k]), k
