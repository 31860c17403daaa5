"""Small helpers mirrored from the reference's ``utils.py`` that the hot path's callers use."""
import torch


def toNP(x):
    """torch tensor -> numpy array on the host (reference utils.py:12-16)."""
    return x.detach().to(torch.device("cpu")).numpy()


def nll_loss(log_probs, labels):
    """Drop-in for ``torch.nn.functional.nll_loss(log_probs, labels)`` (mean reduction, rows x classes) on the HIP path."""
    from . import ops
    return ops.NllLossFn.apply(log_probs, labels)
