"""``diffusion_net.utils`` -- every public name of the reference's ``utils.py`` (utils.py:12-119), because the experiment scripts
either side of the hot path call them on the package: ``random_rotate_points`` (human_segmentation_original.py:124,
classification_shrec11.py:134, rna_mesh_segmentation.py:123), ``ensure_dir_exists`` (human_segmentation_original_dataset.py:142),
``label_smoothing_log_loss`` (classification_shrec11.py:147), ``toNP`` (all datasets), and the precompute's own helpers
(``sparse_np_to_torch`` / ``sparse_torch_to_np`` / ``hash_arrays``).  Same names, arguments and results; own code.
``tests/test_dropin_surface.py`` scans the BASELINE scripts for every ``diffusion_net.<module>.<name>`` they use and checks that
it resolves here."""
import hashlib
import os

import numpy as np
import torch


def toNP(x):
    """torch tensor -> numpy array on the host (reference utils.py:12-16)."""
    return x.detach().to(torch.device("cpu")).numpy()


def nll_loss(log_probs, labels):
    """Drop-in for ``torch.nn.functional.nll_loss(log_probs, labels)`` (mean reduction, rows x classes) on the HIP path: one pass over
    the log-probabilities forward, one backward (dn_head.hip).  Rows labelled -100 (torch's default ``ignore_index``) neither
    contribute nor count; any other label outside [0, C) makes the loss NaN (torch raises a device-side assert there; a NaN is as loud
    and needs no host synchronisation).  Class weights and other reductions are not implemented (use torch); more than 2048 classes
    fall back to ``torch.nn.functional.nll_loss``."""
    from . import _hip, ops
    if log_probs.dim() != 2:
        raise ValueError("nll_loss expects [rows, classes] log-probabilities")
    if log_probs.shape[1] > _hip.HEAD_MAX_CLASSES:
        return torch.nn.functional.nll_loss(log_probs, labels)
    return ops.HeadFn.apply(log_probs, None, labels, False, 0.0, False)[1]


def label_smoothing_log_loss(pred, labels, smoothing=0.0):
    """Reference ``utils.label_smoothing_log_loss`` (utils.py:18-24): cross entropy of log-probabilities against the
    smoothed one-hot target, mean over rows.  The reference builds its one-hot with ``one_hot[labels] = 1``, which is a
    proper one-hot only for the 1-D prediction of its single caller (classification_shrec11.py: one mesh, scalar label);
    that case is reproduced exactly, and 2-D predictions get the per-row one-hot the formula intends.  On a ROCm device the loss
    and its gradient are one HIP kernel each (the NLL kernel with a smoothed target); host tensors use the torch formula."""
    if pred.is_cuda and pred.dtype == torch.float32 and pred.dim() in (1, 2) and pred.shape[-1] <= 2048:
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


# ----------------------------------------------------------------------------------------------
# augmentation used by the training loops (utils.py:30-47, 78-114)
# ----------------------------------------------------------------------------------------------
def random_rotation_matrix(randgen=None):
    """Uniformly distributed 3x3 rotation (up to the reflection convention of the reference) from three uniform variates --
    Arvo's method (Graphics Gems III, "Fast random rotation matrices"): a random rotation about z followed by the Householder
    reflection that sends the pole to a uniformly distributed point, M = (v v^T - I) Rz(theta) with |v|^2 = 2.
    ``randgen``: a ``numpy.random.RandomState`` for reproducibility (three ``rand()`` draws, in the reference's order: theta,
    phi, z), as in utils.py:78-114 -- a seeded generator gives the reference's matrix bit for bit."""
    if randgen is None:
        randgen = np.random.RandomState()
    u_theta, u_phi, u_z = (float(u) for u in randgen.rand(3))
    theta, phi, z = 2.0 * np.pi * u_theta, 2.0 * np.pi * u_phi, 2.0 * u_z
    rad = np.sqrt(z)
    v = np.array([np.sin(phi) * rad, np.cos(phi) * rad, np.sqrt(2.0 - z)])
    c, s_ = np.cos(theta), np.sin(theta)
    rot_z = np.array([[c, s_, 0.0], [-s_, c, 0.0], [0.0, 0.0, 1.0]])
    return (np.outer(v, v) - np.eye(3)).dot(rot_z)


def random_rotate_points(pts, randgen=None):
    """pts [..., 3] (torch, any device) times a random rotation built on the host (utils.py:30-33)."""
    rot = torch.from_numpy(random_rotation_matrix(randgen)).to(device=pts.device, dtype=pts.dtype)
    return torch.matmul(pts, rot)


def random_rotate_points_y(pts):
    """Random rotation about the y axis, angle drawn with torch's generator on the points' device (utils.py:35-47)."""
    angle = torch.rand(1, device=pts.device, dtype=pts.dtype) * (2.0 * np.pi)
    c, s_ = torch.cos(angle), torch.sin(angle)
    rot = torch.zeros(3, 3, device=pts.device, dtype=pts.dtype)
    rot[0, 0], rot[0, 2], rot[2, 0], rot[2, 2], rot[1, 1] = c, s_, -s_, c, 1.0
    return torch.matmul(pts, rot)


# ----------------------------------------------------------------------------------------------
# scipy <-> torch sparse, cache keys, directories (utils.py:50-76, 117-119)
# ----------------------------------------------------------------------------------------------
def sparse_np_to_torch(A):
    """scipy sparse matrix -> coalesced torch sparse COO, int64 indices, fp32 values (utils.py:50-55)."""
    coo = A.tocoo()
    idx = torch.from_numpy(np.vstack((coo.row, coo.col)).astype(np.int64))
    val = torch.from_numpy(np.asarray(coo.data)).to(torch.float32)
    return torch.sparse_coo_tensor(idx, val, torch.Size(coo.shape)).coalesce()


def sparse_torch_to_np(A):
    """coalesced 2-D torch sparse COO -> scipy CSC (utils.py:58-67)."""
    import scipy.sparse
    if len(A.shape) != 2:
        raise RuntimeError("should be a matrix-shaped type; dim is : " + str(A.shape))
    return scipy.sparse.coo_matrix((toNP(A.values()), toNP(A.indices())), shape=A.shape).tocsc()


def hash_arrays(arrs):
    """sha1 hex digest over the raw bytes of the arrays in order -- the reference's operator-cache key (utils.py:71-76)."""
    h = hashlib.sha1()
    for a in arrs:
        h.update(np.ascontiguousarray(a).view(np.uint8))
    return h.hexdigest()


def ensure_dir_exists(d):
    """utils.py:117-119"""
    if not os.path.exists(d):
        os.makedirs(d)
