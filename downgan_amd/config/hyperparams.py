"""Hyper-parameters of the hot path: same names and defaults as the reference's module constants
(reference DoWnGAN/config/hyperparams.py:16-25; Adam betas from DoWnGAN/GAN/stage.py:63-64)."""
gp_lambda = 10
critic_iterations = 5
batch_size = 32
gamma = 0.01
content_lambda = 5
lr = 0.00025
betas = (0.9, 0.99)
epochs = 1000


def as_engine_hp(batch=None):
    from ..engine import HyperParams
    return HyperParams(gp_lambda=float(gp_lambda), critic_iterations=critic_iterations,
                       batch_size=batch or batch_size, gamma=gamma, content_lambda=float(content_lambda),
                       lr=lr, beta1=betas[0], beta2=betas[1])


def __getattr__(name):
    # reference hyperparams.py:2-7,38-43: the metric table binds the loss functions; resolved lazily here so that importing
    # the constants does not import the compute backend
    if name == "metrics_to_calculate":
        from ..GAN.losses import metrics_to_calculate
        return metrics_to_calculate
    raise AttributeError(name)
