"""Small helpers mirrored from the reference's ``utils.py`` that the hot path's callers use."""
import torch


def toNP(x):
    """torch tensor -> numpy array on the host (reference utils.py:12-16)."""
    return x.detach().to(torch.device("cpu")).numpy()


def nll_loss(log_probs, labels):
    """Drop-in for ``torch.nn.functional.nll_loss(log_probs, labels)`` (mean reduction, rows x classes) on the HIP path: one pass over
    the log-probabilities forward, one backward (dn_head.hip).  Rows whose label lies outside [0, C) -- e.g. ``ignore_index=-100`` --
    neither contribute nor count, as in torch; class weights and other reductions are not implemented (TypeError / use torch)."""
    from . import ops
    if log_probs.dim() != 2:
        raise ValueError("nll_loss expects [rows, classes] log-probabilities")
    return ops.HeadFn.apply(log_probs, None, labels, False, 0.0, False)[1]


def label_smoothing_log_loss(pred, labels, smoothing=0.0):
    """Reference ``utils.label_smoothing_log_loss`` (utils.py:18-24): cross entropy of log-probabilities against the
    smoothed one-hot target, mean over rows.  The reference builds its one-hot with ``one_hot[labels] = 1``, which is a
    proper one-hot only for the 1-D prediction of its single caller (classification_shrec11.py: one mesh, scalar label);
    that case is reproduced exactly, and 2-D predictions get the per-row one-hot the formula intends.  On a ROCm device the loss
    and its gradient are one HIP kernel each (the NLL kernel with a smoothed target); host tensors use the torch formula."""
    if pred.is_cuda and pred.dtype == torch.float32 and pred.dim() in (1, 2):
        from . import ops
        p2 = pred.reshape(1, -1) if pred.dim() == 1 else pred
        lab = labels.reshape(-1).to(torch.int64)
        return ops.HeadFn.apply(p2, None, lab, False, float(smoothing), False)[1]
    n_class = pred.shape[-1]
    one_hot = torch.zeros_like(pred)
    if pred.dim() == 1:
        one_hot[labels] = 1.0
    else:
        one_hot.scatter_(-1, labels.reshape(*pred.shape[:-1], 1).long(), 1.0)
    one_hot = one_hot * (1 - smoothing) + (1 - one_hot) * smoothing / (n_class - 1)
    return -(one_hot * pred).sum(dim=-1).mean()
