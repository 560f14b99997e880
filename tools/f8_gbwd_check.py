"""Diagnostic: generator parameter gradients, HipOps vs EmuOps, with / without the fp8 dense-block data gradients (f8_gbwd).
Prints rel-l2 per parameter group: (hip, emu) with f8_gbwd; (hip, emu) without; (emu with, emu without) = the format's own error."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from downgan_amd import synthetic
from downgan_amd.engine import NativeGenerator
from downgan_amd.layout import nchw_to_nhwc_padded
from downgan_amd.ops import HipOps
from oracle.emu_ops import EmuOps

torch.set_num_threads(8)
B, S, F_, cin, nrb = 2, 16, 128, 2, 2
pg = synthetic.generator_params(F_, cin, 2, nrb)
coarse, _ = synthetic.tiles(B, cin, S)
res = {}
for gb in (True, False):
    emu_ops = EmuOps("bf16", f8_generator=True); emu_ops.f8_gbwd = gb
    emu = NativeGenerator(emu_ops, F_, cin, B, S, num_res_blocks=nrb)
    emu.load_state_dict(pg)
    xe = nchw_to_nhwc_padded(torch.from_numpy(coarse), 16, torch.bfloat16)
    fake = emu.forward(xe, save=True)
    dfake = (torch.randn(fake.shape, generator=torch.Generator().manual_seed(5)) * (torch.arange(fake.shape[-1]) < 2)).to(torch.bfloat16)
    emu.P.zero_grad(); emu.backward(xe, dfake)
    ops = HipOps("bf16", f8_generator=True); ops.f8_gbwd = gb
    G = NativeGenerator(ops, F_, cin, B, S, num_res_blocks=nrb)
    G.load_state_dict(pg)
    xc = ops.zeros(B, S, S, 16); ops.nchw_to_nhwc(torch.from_numpy(coarse).cuda(), xc)
    G.forward(xc, save=True)
    G.P.zero_grad(); G.backward(xc, dfake.cuda())
    res[gb] = (emu.grad_dict(), {k: v.float().cpu() for k, v in G.grad_dict().items()})
rel = lambda a, b: float((a - b).norm() / max(1e-12, float(a.norm())))
print(f"{'parameter':48s} hip-emu(f8)  hip-emu(bf16)  emu f8-bf16   hip f8-bf16")
for name in res[True][0]:
    if not name.endswith("weight"):
        continue
    e8, h8 = res[True][0][name], res[True][1][name]
    e16, h16 = res[False][0][name], res[False][1][name]
    print(f"{name:48s} {rel(e8, h8):10.4f} {rel(e16, h16):12.4f} {rel(e16, e8):12.4f} {rel(h16, h8):12.4f}")
