"""TEST INFRASTRUCTURE — CPU restatement (PyTorch-CPU fp32 ops) of the DoWnGAN WGAN-GP hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``downgan_amd``) never does.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks this restatement against golden
vectors captured from the real reference (``tests/golden/make_golden.py`` imports
``/root/reference`` in the build container; the fixtures it wrote are committed).

Each function cites the reference lines it restates (paths relative to the reference root).
The networks are restated functionally over a ``{state_dict name: tensor}`` dict so that the
same parameter arrays drive the reference, this oracle and the HIP path.
"""
from __future__ import annotations

import dataclasses
import math

import torch
import torch.nn.functional as F


@dataclasses.dataclass
class HP:
    """DoWnGAN/config/hyperparams.py:16-22 and GAN/stage.py:63-64 (Adam)."""
    gp_lambda: float = 10.0
    critic_iterations: int = 5
    batch_size: int = 32
    gamma: float = 0.01
    content_lambda: float = 5.0
    lr: float = 0.00025
    beta1: float = 0.9
    beta2: float = 0.99
    eps: float = 1e-8


# --------------------------------------------------------------------------- networks
def dense_residual_block(P, prefix, x, res_scale=0.2):
    """networks/generator.py:36-41 (b1..b4 conv+LeakyReLU(0.01), b5 conv only; cat; 0.2*out + x)."""
    inputs = x
    out = None
    for k in range(1, 6):
        out = F.conv2d(inputs, P[f"{prefix}.b{k}.0.weight"], P[f"{prefix}.b{k}.0.bias"], stride=1, padding=1)
        if k < 5:
            out = F.leaky_relu(out, 0.01)  # nn.LeakyReLU() default slope, generator.py:26
        inputs = torch.cat([inputs, out], 1)
    return out.mul(res_scale) + x


def rrdb(P, prefix, x, res_scale=0.2):
    """networks/generator.py:44-53."""
    out = x
    for j in range(3):
        out = dense_residual_block(P, f"{prefix}.dense_blocks.{j}", out)
    return out.mul(res_scale) + x


def generator_forward(P, x, num_res_blocks=16, num_upsample=3):
    """networks/generator.py:83-90."""
    out1 = F.conv2d(x, P["conv1.weight"], P["conv1.bias"], padding=1)
    out = out1
    for i in range(num_res_blocks):
        out = rrdb(P, f"res_blocks.{i}", out)
    out2 = F.conv2d(out, P["conv2.weight"], P["conv2.bias"], padding=1)
    out = torch.add(out1, out2)
    for u in range(num_upsample):
        out = F.conv2d(out, P[f"upsampling.{3 * u}.weight"], P[f"upsampling.{3 * u}.bias"], padding=1)
        out = F.leaky_relu(out, 0.01)
        out = F.pixel_shuffle(out, 2)
    out = F.conv2d(out, P["conv3.0.weight"], P["conv3.0.bias"], padding=1)
    out = F.leaky_relu(out, 0.01)
    out = F.conv2d(out, P["conv3.2.weight"], P["conv3.2.bias"], padding=1)
    return out


CRITIC_STRIDES = (1, 2, 1, 2, 1, 2, 1, 2)  # networks/critic.py:20-88


def critic_forward(P, x):
    """networks/critic.py:101-106: 8 conv3x3 + LeakyReLU(0.2), NCHW flatten, Linear-LeakyReLU-Linear."""
    out = x
    for li, st in enumerate(CRITIC_STRIDES):
        b = P.get(f"features.{2 * li}.bias")
        out = F.conv2d(out, P[f"features.{2 * li}.weight"], b, stride=st, padding=1)
        out = F.leaky_relu(out, 0.2)
    out = torch.flatten(out, 1)
    out = F.linear(out, P["classifier.0.weight"], P["classifier.0.bias"])
    out = F.leaky_relu(out, 0.2)
    out = F.linear(out, P["classifier.2.weight"], P["classifier.2.bias"])
    return out


# --------------------------------------------------------------------------- losses
def gradient_penalty(PC, real, fake, alpha, hp: HP):
    """GAN/wasserstein.py:87-117.  ``alpha`` [B] replaces the device RNG draw at :91.

    Returns ``gp_lambda * mean((||grad||-1)^2)`` with a graph (create_graph) like :100-117.
    The reference reshapes with the global ``hp.batch_size`` (:110); callers keep
    ``hp.batch_size == real.size(0)`` (full batches only).
    """
    a = alpha.view(-1, 1, 1, 1).expand_as(real)
    interpolated = (a * real.detach() + (1 - a) * fake.detach()).requires_grad_(True)
    critic_interpolated = critic_forward(PC, interpolated)
    gradients = torch.autograd.grad(outputs=critic_interpolated, inputs=interpolated,
                                    grad_outputs=torch.ones_like(critic_interpolated),
                                    create_graph=True, retain_graph=True)[0]
    gradients = gradients.view(hp.batch_size, -1)
    gradients_norm = torch.sqrt(torch.sum(gradients ** 2, dim=1) + 1e-12)
    return hp.gp_lambda * ((gradients_norm - 1) ** 2).mean()


def content_loss(hr, fake):
    """GAN/losses.py:40-55: nn.L1Loss() (mean absolute error)."""
    return F.l1_loss(hr, fake)


# --------------------------------------------------------------------------- optimiser
class Adam:
    """torch.optim.Adam(lr, betas=(0.9,0.99)) as built at GAN/stage.py:63-64 (eps 1e-8, no decay)."""

    def __init__(self, params: dict, hp: HP):
        self.hp = hp
        self.t = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    @torch.no_grad()
    def step(self, params: dict, grads: dict):
        hp = self.hp
        self.t += 1
        bc1 = 1 - hp.beta1 ** self.t
        bc2 = 1 - hp.beta2 ** self.t
        for k, p in params.items():
            g = grads.get(k)
            if g is None:
                continue
            self.m[k].mul_(hp.beta1).add_(g, alpha=1 - hp.beta1)
            self.v[k].mul_(hp.beta2).addcmul_(g, g, value=1 - hp.beta2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(hp.eps)
            p.addcdiv_(self.m[k], denom, value=-hp.lr / bc1)


# --------------------------------------------------------------------------- the train step
class OracleTrainer:
    """Restates WassersteinGAN (GAN/wasserstein.py:16-189) over parameter dicts.

    Unlike the reference, the iteration methods RETURN the scalars the reference computes and
    drops (:46-50, :74-78) and the parameter gradients.
    """

    def __init__(self, PG: dict, PC: dict, hp: HP = None, num_res_blocks=16, num_upsample=3):
        self.hp = hp or HP()
        self.PG = {k: v.clone().requires_grad_(True) for k, v in PG.items()}
        self.PC = {k: v.clone().requires_grad_(True) for k, v in PC.items()}
        self.nrb, self.nup = num_res_blocks, num_upsample
        self.G_opt = Adam(self.PG, self.hp)
        self.C_opt = Adam(self.PC, self.hp)
        self.num_steps = 0

    def G(self, x):
        return generator_forward(self.PG, x, self.nrb, self.nup)

    def C(self, x):
        return critic_forward(self.PC, x)

    def critic_iteration(self, coarse, fine, alpha, apply_update=True):
        """GAN/wasserstein.py:27-55.  ``fake`` is computed without a graph into G: the reference's
        G backward here is dead work (G grads are zeroed at :65 before any use)."""
        hp = self.hp
        with torch.no_grad():
            fake = self.G(coarse)
        c_real = self.C(fine)
        c_fake = self.C(fake)
        gp_ret = gradient_penalty(self.PC, fine, fake, alpha, hp)
        gradient_penalty_term = hp.gp_lambda * gp_ret  # :40 (second multiply; effective weight lambda^2)
        c_real_mean = torch.mean(c_real)
        c_fake_mean = torch.mean(c_fake)
        critic_loss = c_fake_mean - c_real_mean + gradient_penalty_term
        w_estimate = c_real_mean - c_fake_mean
        names = list(self.PC)
        gl = torch.autograd.grad(critic_loss, [self.PC[k] for k in names], allow_unused=True)
        grads = {k: (g if g is not None else torch.zeros_like(self.PC[k])) for k, g in zip(names, gl)}
        if apply_update:
            self.C_opt.step(self.PC, grads)
        return {"c_real_mean": c_real_mean.item(), "c_fake_mean": c_fake_mean.item(),
                "gp_ret": gp_ret.item(), "gradient_penalty": gradient_penalty_term.item(),
                "critic_loss": critic_loss.item(), "w_estimate": w_estimate.item()}, grads

    def generator_iteration(self, coarse, fine, apply_update=True):
        """GAN/wasserstein.py:58-83."""
        hp = self.hp
        fake = self.G(coarse)
        c_fake = self.C(fake)
        g_adv = -torch.mean(c_fake) * hp.gamma
        cl = content_loss(fake, fine)
        g_loss = g_adv + hp.content_lambda * cl
        names = list(self.PG)
        gl = torch.autograd.grad(g_loss, [self.PG[k] for k in names])
        grads = dict(zip(names, gl))
        if apply_update:
            self.G_opt.step(self.PG, grads)
        return {"g_loss": g_loss.item(), "content_loss": cl.item(), "g_c_fake_mean": torch.mean(c_fake).item()}, grads

    @torch.no_grad()
    def metrics(self, coarse, fine):
        """mlflow_tools/mlflow_epoch.py:53-63 restricted to the metrics whose arithmetic is in the reference itself:
        MAE (losses.py:51-53), MSE (losses.py:68-69), Wass (losses.py:8-9).  MS-SSIM goes through the unpinned
        third-party pytorch_msssim and is not restated."""
        fake = self.G(coarse)
        creal = torch.mean(self.C(fine))
        cfake = torch.mean(self.C(fake))
        return {"MAE": F.l1_loss(fine, fake).item(), "MSE": F.mse_loss(fine, fake).item(), "Wass": (creal - cfake).item()}

    def train_step(self, coarse, fine, alpha):
        """Loop body of _train_epoch, GAN/wasserstein.py:131-147 (metrics pass :140 excluded)."""
        out, _ = self.critic_iteration(coarse, fine, alpha)
        if self.num_steps % self.hp.critic_iterations == 0:
            g, _ = self.generator_iteration(coarse, fine)
            out.update(g)
        self.num_steps += 1
        return out


# --------------------------------------------------------------------------- frequency-separation variant
def lowpass(x):
    """config/hyperparams.py:31-35: ``low(rf(x))`` = AvgPool2d(5, stride=1, padding=0) after ReplicationPad2d(2)."""
    return F.avg_pool2d(F.pad(x, (2, 2, 2, 2), mode="replicate"), 5, stride=1, padding=0)


class OracleTrainerFS(OracleTrainer):
    """Restates WassersteinGANFS (GAN/wasserstein_fs.py:28-130): the critic sees the high-pass parts ``x - low(x)`` of the
    real and generated fields (also inside the gradient penalty), the content loss compares the low-pass parts.

    Parity status of this variant: UNPINNED.  The reference module cannot be imported (it imports ``DoWnGAN.gen_grid_plots``,
    a bare ``hyperparams`` and an undefined ``config``, wasserstein_fs.py:2-10, and no caller reads ``hp.freq_sep``), so no
    golden can be generated from it; this follows the text of its two iteration methods."""

    def critic_iteration(self, coarse, fine, alpha, apply_update=True):
        """wasserstein_fs.py:28-60."""
        hp = self.hp
        with torch.no_grad():
            fake = self.G(coarse)
            fake_high = fake - lowpass(fake)
            real_high = fine - lowpass(fine)
        c_real = self.C(real_high)
        c_fake = self.C(fake_high)
        gp_ret = gradient_penalty(self.PC, real_high, fake_high, alpha, hp)
        gradient_penalty_term = hp.gp_lambda * gp_ret                     # :45
        c_real_mean, c_fake_mean = torch.mean(c_real), torch.mean(c_fake)
        critic_loss = c_fake_mean - c_real_mean + gradient_penalty_term
        names = list(self.PC)
        gl = torch.autograd.grad(critic_loss, [self.PC[k] for k in names], allow_unused=True)
        grads = {k: (g if g is not None else torch.zeros_like(self.PC[k])) for k, g in zip(names, gl)}
        if apply_update:
            self.C_opt.step(self.PC, grads)
        return {"c_real_mean": c_real_mean.item(), "c_fake_mean": c_fake_mean.item(), "gp_ret": gp_ret.item(),
                "gradient_penalty": gradient_penalty_term.item(), "critic_loss": critic_loss.item(),
                "w_estimate": (c_real_mean - c_fake_mean).item()}, grads

    def generator_iteration(self, coarse, fine, apply_update=True):
        """wasserstein_fs.py:63-92."""
        hp = self.hp
        fake = self.G(coarse)
        fake_low = lowpass(fake)
        real_low = lowpass(fine)
        fake_high = fake - fake_low
        c_fake = self.C(fake_high)
        cl = content_loss(fake_low, real_low)                             # :86
        g_loss = -torch.mean(c_fake) * hp.gamma + hp.content_lambda * cl
        names = list(self.PG)
        gl = torch.autograd.grad(g_loss, [self.PG[k] for k in names])
        grads = dict(zip(names, gl))
        if apply_update:
            self.G_opt.step(self.PG, grads)
        return {"g_loss": g_loss.item(), "content_loss": cl.item(), "g_c_fake_mean": torch.mean(c_fake).item()}, grads
