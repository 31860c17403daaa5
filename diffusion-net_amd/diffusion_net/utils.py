"""Small helpers mirrored from the reference's ``utils.py`` that the hot path's callers use."""
import torch


def toNP(x):
    """torch tensor -> numpy array on the host (reference utils.py:12-16)."""
    return x.detach().to(torch.device("cpu")).numpy()
