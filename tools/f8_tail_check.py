import sys; sys.path.insert(0, '/root/repo')
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
xe = nchw_to_nhwc_padded(torch.from_numpy(coarse), 16, torch.bfloat16)
def emu_run(f8, tail):
    o = EmuOps("bf16", f8_generator=f8); o.f8_gtail = tail
    G = NativeGenerator(o, F_, cin, B, S, num_res_blocks=nrb); G.load_state_dict(pg)
    return G.forward(xe, save=True).float()[..., :2], G.trunk.float().clone()
def hip_run(f8, tail):
    o = HipOps("bf16", f8_generator=f8); o.f8_gtail = tail
    G = NativeGenerator(o, F_, cin, B, S, num_res_blocks=nrb); G.load_state_dict(pg)
    xc = o.zeros(B, S, S, 16); o.nchw_to_nhwc(torch.from_numpy(coarse).cuda(), xc)
    return G.forward(xc, save=True).float().cpu()[..., :2], G.trunk.float().cpu()
e16, t16 = emu_run(False, False)
res = {}
for tail in (False, True):
    ef, et = emu_run(True, tail); hf, ht = hip_run(True, tail)
    mx = lambda a, b: float((a - b).abs().max() / b.abs().max())
    l2 = lambda a, b: float((a - b).norm() / b.norm())
    print(f"tail={tail}: fake hip-emu max {mx(hf, ef):.4f} l2 {l2(hf, ef):.4f} | emu f8-bf16 max {mx(ef, e16):.4f} l2 {l2(ef, e16):.4f} | hip f8 - emu bf16 max {mx(hf, e16):.4f} l2 {l2(hf, e16):.4f} | trunk hip-emu max {mx(ht, et):.4f} l2 {l2(ht, et):.4f}; trunk emu f8-bf16 l2 {l2(et, t16):.4f}")
h16, _ = hip_run(False, False)
print(f"bf16: fake hip-emu max {mx(h16, e16):.4f} l2 {l2(h16, e16):.4f}")
