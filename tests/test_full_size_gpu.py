"""Size-independent properties at BASELINE.json configs[1] FULL launch sizes (batch 32, 1024x1024 tiles: tensors of up to 8.6 GB,
i.e. byte offsets beyond 2^32 and grids of 131 072 workgroups), where the oracle is out of reach:

* batch replication: a batch of 32 copies of one image must give 32 identical outputs, equal to the batch-1 result
  (catches any addressing error at high offsets / large grids), for forward and data gradient;
* linearity in the batch: the weight gradient of the replicated batch is 32x the batch-1 gradient (fp32 accumulation order
  differs: relative 1e-4);
on the layers with the largest tensors of the step -- the critic's first three convs (critic.py:21-42; im2col kernel, the stride-2
halo kernel, the 256-channel stride-1 layer) and the generator's last 128-channel conv at 1024^2 (generator.py:77) -- in bf16 and,
for the critic's wide layers, on the fp8 kernel."""
import pytest
import torch

from downgan_amd.ops import Conv, HipOps

pytestmark = pytest.mark.gpu
B = 32

LAYERS = [
    # name, H, Cin, Cout, stride, cin_real, net, modes
    ("critic features.0", 1024, 16, 128, 1, 2, "C", ("bf16",)),
    ("critic features.2", 1024, 128, 128, 2, 0, "C", ("bf16", "fp8")),
    ("critic features.4", 512, 128, 256, 1, 0, "C", ("bf16", "fp8")),
    ("critic features.6", 512, 256, 256, 2, 0, "C", ("bf16", "fp8")),      # 8-wave stride-2 tile
    ("generator conv3.0", 1024, 128, 128, 1, 0, "", ("bf16",)),
]


@pytest.mark.parametrize("layer", LAYERS, ids=[l[0].replace(" ", "_") for l in LAYERS])
def test_batch_replication_and_linearity_at_full_size(layer):
    name, H, ci, co, st, cin_real, net, modes = layer
    g = torch.Generator().manual_seed(100 + [l[0] for l in LAYERS].index(name))
    x1 = torch.randn(1, H, H, ci, generator=g).to(torch.bfloat16)
    if cin_real:
        x1[..., cin_real:] = 0
    w = (torch.randn(co * 9 * ci, generator=g) * 0.05).to(torch.bfloat16).cuda()
    dy1 = torch.randn(1, H // st, H // st, co, generator=g).to(torch.bfloat16).cuda()
    x1 = x1.cuda()
    for mode in modes:
        ops = HipOps("bf16", f8_critic=mode == "fp8")
        cv1 = Conv(1, H, H, ci, co, st, False, cin_real=cin_real, net=net)
        cvB = Conv(B, H, H, ci, co, st, False, cin_real=cin_real, net=net)
        # forward
        y1 = ops.zeros(*ops.out_shape(cv1))
        ops.conv_fwd(cv1, x1, w, y1, act=0.2)
        xB = x1.expand(B, H, H, ci).contiguous()
        yB = ops.zeros(*ops.out_shape(cvB))
        ops.conv_fwd(cvB, xB, w, yB, act=0.2)
        assert ops.lib.dg_last_conv_kernels() == (32 if mode == "fp8" else 16 if cin_real else 8)
        assert float(y1.float().abs().max()) > 0
        for b in (0, 1, 15, 30, 31):
            assert torch.equal(yB[b], y1[0]), (name, mode, "fwd", b)
        assert torch.equal(yB, y1.expand_as(yB)), (name, mode, "fwd")
        del yB
        # weight gradient (bf16 kernels in every mode): linear in the batch
        dw1 = torch.zeros(co * 9 * ci, dtype=torch.float32).cuda()
        ops.conv_wgrad(cv1, x1, dy1, dw1)
        dyB = dy1.expand(B, H // st, H // st, co).contiguous()
        dwB = torch.zeros(co * 9 * ci, dtype=torch.float32).cuda()
        ops.conv_wgrad(cvB, xB, dyB, dwB)
        err = float((dwB - B * dw1).abs().max()) / float((B * dw1).abs().max())
        assert err < 1e-4, (name, mode, "wgrad", err)
        del xB
        # data gradient (not for the 2-channel layer: its input gradient is a 16-channel-output launch of another kernel family,
        # covered at its full size by the GP pass of test_cfg2_parity_gpu.py)
        if not cin_real:
            dx1 = ops.zeros(1, H, H, ci)
            ops.conv_dgrad(cv1, dy1, w, dx1)
            dxB = ops.zeros(B, H, H, ci)
            ops.conv_dgrad(cvB, dyB, w, dxB)
            assert float(dx1.float().abs().max()) > 0
            assert torch.equal(dxB, dx1.expand_as(dxB)), (name, mode, "dgrad")
            del dxB
        del dyB
        torch.cuda.empty_cache()
