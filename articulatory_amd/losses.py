"""GAN losses of the reference's train step on the native discriminators' outputs (``HiFiGANMultiScaleMultiPeriodDiscriminator(x,
native=True)``): the same values as articulatory/losses/adversarial_loss.py:12-123 and feat_match_loss.py:12-54 on the reference-shaped
tensors — every term is a mean over the elements of a layer output, which does not depend on how the engine lays them out.  Plain torch
element-wise reductions on views of the engine's buffers (differentiable: the gradients land in the buffers' own layout)."""
import torch


def _mean(out, fn):
    """mean over the valid elements of one DiscOutput of fn(view)."""
    return sum(fn(v).sum() for v in out.valid()) / out.numel()


def _last(o):
    return o[-1] if isinstance(o, (list, tuple)) else o


def generator_adversarial_loss(outputs, average_by_discriminators=True, loss_type="mse"):
    """adversarial_loss.py:12-62."""
    assert loss_type in ("mse", "hinge")
    loss = 0.0
    for i, o in enumerate(outputs):
        o = _last(o)
        loss = loss + (_mean(o, lambda v: (v - 1.0) ** 2) if loss_type == "mse" else -_mean(o, lambda v: v))
    return loss / (i + 1) if average_by_discriminators else loss


def discriminator_adversarial_loss(outputs_hat, outputs, average_by_discriminators=True, loss_type="mse"):
    """adversarial_loss.py:65-123 -> (real_loss, fake_loss)."""
    assert loss_type in ("mse", "hinge")
    real, fake = 0.0, 0.0
    for i, (oh, o) in enumerate(zip(outputs_hat, outputs)):
        oh, o = _last(oh), _last(o)
        if loss_type == "mse":
            real = real + _mean(o, lambda v: (v - 1.0) ** 2)
            fake = fake + _mean(oh, lambda v: v ** 2)
        else:
            real = real - _mean(o, lambda v: torch.clamp(v - 1.0, max=0.0))
            fake = fake - _mean(oh, lambda v: torch.clamp(-v - 1.0, max=0.0))
    if average_by_discriminators:
        real, fake = real / (i + 1), fake / (i + 1)
    return real, fake


def feature_match_loss(feats_hat, feats, average_by_layers=True, average_by_discriminators=True, include_final_outputs=False):
    """feat_match_loss.py:12-54 (the ground-truth features are constants)."""
    total = 0.0
    for i, (fh, f) in enumerate(zip(feats_hat, feats)):
        if not include_final_outputs:
            fh, f = fh[:-1], f[:-1]
        part = 0.0
        for j, (a, b) in enumerate(zip(fh, f)):
            part = part + sum((va - vb.detach()).abs().sum() for va, vb in zip(a.valid(), b.valid())) / a.numel()
        if average_by_layers:
            part = part / (j + 1)
        total = total + part
    return total / (i + 1) if average_by_discriminators else total
