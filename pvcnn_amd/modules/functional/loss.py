"""kl_loss / huber_loss (reference: modules/functional/loss.py:7-17) -- plain torch, tiny tensors."""
import torch
import torch.nn.functional as tf

__all__ = ['kl_loss', 'huber_loss']


def kl_loss(x, y):
    """KL(softmax(x).detach() || softmax(y)) averaged over all but the class dimension (dim 1)."""
    p = tf.softmax(x.detach(), dim=1)
    log_q = tf.log_softmax(y, dim=1)
    return (p * (p.log() - log_q)).sum(dim=1).mean()


def huber_loss(error, delta):
    """Mean Huber loss: 0.5*e^2 for |e| <= delta, delta*(|e| - 0.5*delta) beyond."""
    mag = error.abs()
    inner = mag.clamp(max=delta)
    return (0.5 * inner * inner + delta * (mag - inner)).mean()
