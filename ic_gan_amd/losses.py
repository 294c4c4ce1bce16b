"""GAN losses of the IC-GAN step (reference BigGAN_PyTorch/losses.py:12-43).  The operands are the
[B,1] discriminator logits: a handful of scalars per step, kept as PyTorch expressions on the device."""
import torch
import torch.nn.functional as F


def loss_dcgan_dis(dis_fake, dis_real):
    return torch.mean(F.softplus(-dis_real)), torch.mean(F.softplus(dis_fake))


def loss_dcgan_gen(dis_fake):
    return torch.mean(F.softplus(-dis_fake))


def loss_hinge_dis(dis_fake, dis_real):
    """-> (loss_real, loss_fake)"""
    return torch.mean(F.relu(1.0 - dis_real)), torch.mean(F.relu(1.0 + dis_fake))


def loss_hinge_gen(dis_fake):
    return -torch.mean(dis_fake)


generator_loss = loss_hinge_gen
discriminator_loss = loss_hinge_dis
