"""Point-set metrics and export for generated point clouds on MI355X.

Same function names, arguments and results as the reference's evaluation code
  compute_chamfer_distance, compute_emd_distance            test_optimize.py:354-415
  distChamfer, emd_approx, robust_chamfer_distance, robust_emd   train_newloss.py:316-385
  GlobalNormalizer                                           test_optimize.py:32-77
  np.save of each generated point cloud                      README.md:108-113
with the O(N M) distance work (torch.cdist + min, the assignment cost matrix) in libnova_hip.so
(csrc/pointset.hip). The optimal assignment itself is scipy's linear_sum_assignment on the host, exactly as in the
reference. GPU tensors only: there is no CPU path here (NovaHipError for CPU tensors or a missing library).
"""
import json
import os

import numpy as np
import torch

from . import hip


def _points(t, name):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise hip.NovaHipError(f"{name}: point-set metrics run on the GPU (got {'a CPU tensor' if torch.is_tensor(t) else type(t).__name__})")
    if t.dim() != 3 or t.shape[-1] != 3:
        raise ValueError(f"{name}: expected [B, n, 3] points, got {tuple(t.shape)}")
    if t.requires_grad and torch.is_grad_enabled():
        # evaluation metrics: the HIP distance kernels have no backward, so a loss built on them would silently carry no gradient
        raise hip.NovaHipError(f"{name}: nova_pointcloud_amd.metrics is evaluation-only (no autograd through the HIP kernels); "
                               "detach the points or compute a training loss in PyTorch")
    return t.detach().float().contiguous()


def nn_dist(x, y, clamp, unit_norm=False):
    """d[b, i] = min_j ||x[b, i] - y[b, j]|| after clamping coordinates to [-clamp, clamp] (optionally unit-normalised)."""
    x, y = _points(x, "x"), _points(y, "y")
    B, N, M = x.shape[0], x.shape[1], y.shape[1]
    d = torch.empty(B, N, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        hip.call("nova_pointset_nn_dist", x.data_ptr(), y.data_ptr(), d.data_ptr(), B, N, M, -float(clamp), float(clamp),
                 1 if unit_norm else 0, hip.stream_ptr())
    return d


def pairwise_dist(x, y, clamp):
    """D[b, i, j] = ||x[b, i] - y[b, j]|| (the `torch.cdist` cost matrix of the EMD assignment)."""
    x, y = _points(x, "x"), _points(y, "y")
    B, N, M = x.shape[0], x.shape[1], y.shape[1]
    D = torch.empty(B, N, M, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        hip.call("nova_pointset_pairwise_dist", x.data_ptr(), y.data_ptr(), D.data_ptr(), B, N, M, -float(clamp), float(clamp),
                 hip.stream_ptr())
    return D


# ----------------------------------------------------------------------------------------------------
# test_optimize.py:354-415
# ----------------------------------------------------------------------------------------------------
def compute_chamfer_distance(pred, target):
    """Density-weighted Chamfer distance (test_optimize.py:354-381): coordinates clamped to +-5, both sets cut to the
    smaller point count, nearest-neighbour distances weighted by 1 / (d.detach() + 1e-6), result clamped to [0, 10]."""
    n = min(pred.shape[1], target.shape[1])
    pred, target = pred[:, :n], target[:, :n]
    d_pt, d_tp = nn_dist(pred, target, 5.0), nn_dist(target, pred, 5.0)
    dist1 = (d_pt * (1.0 / (d_pt + 1e-6))).mean(dim=1)
    dist2 = (d_tp * (1.0 / (d_tp + 1e-6))).mean(dim=1)
    return (dist1 + dist2).mean().clamp(0.0, 10.0)


def _assignment_mean(cost):
    from scipy.optimize import linear_sum_assignment

    rows, cols = linear_sum_assignment(cost)
    return cost[rows, cols].mean()


def compute_emd_distance(pred, target):
    """Earth mover's distance by optimal assignment (test_optimize.py:385-415): clamp +-5, equal point counts, mean
    matched distance per sample (scipy linear_sum_assignment on the host, as in the reference), batch mean in [0, 10]."""
    n = min(pred.shape[1], target.shape[1])
    cost = pairwise_dist(pred[:, :n], target[:, :n], 5.0).cpu().numpy()
    emd = torch.tensor([float(_assignment_mean(c)) for c in cost], dtype=torch.float32, device=pred.device)
    return emd.mean().clamp(0.0, 10.0)


# ----------------------------------------------------------------------------------------------------
# train_newloss.py:316-385
# ----------------------------------------------------------------------------------------------------
def distChamfer(a, b):
    """(dl, dr) of train_newloss.py:316-349: points clamped to +-1 and scaled to unit norm, nearest-neighbour distance
    floored at 1e-8, passed through log(d + 1e-8) clamped to [-10, 10] and exp (min and the monotone maps commute)."""
    through_log = lambda d: torch.log(d.clamp_min(1e-8) + 1e-8).clamp(-10, 10).exp().mean()
    return through_log(nn_dist(a, b, 1.0, unit_norm=True)), through_log(nn_dist(b, a, 1.0, unit_norm=True))


def emd_approx(x, y):
    """Per-sample EMD of train_newloss.py:352-372 (clamp +-2, distances floored at 1e-8): tensor [B]."""
    assert x.size(1) == y.size(1), "EMD only works if two point clouds are equal size"
    cost = np.maximum(pairwise_dist(x, y, 2.0).cpu().numpy(), 1e-8)
    return torch.from_numpy(np.stack([_assignment_mean(c) for c in cost]).reshape(-1)).to(x)


def robust_chamfer_distance(pred, gt):
    dl, dr = distChamfer(pred, gt)
    return (dl.mean() + dr.mean()) / 2


def robust_emd(pred, gt):
    return emd_approx(pred, gt).mean()


# ----------------------------------------------------------------------------------------------------
# normalisation and export
# ----------------------------------------------------------------------------------------------------
class GlobalNormalizer(object):
    """(x - mean) / std with the training-set statistics of stats.json (test_optimize.py:32-75); identity statistics when
    the file is missing or unreadable, as in the reference."""

    def __init__(self):
        self.global_mean, self.global_std, self.is_fitted = None, None, False

    def load_stats(self, filepath="stats.json"):
        ok = False
        try:
            with open(filepath) as f:
                stats = json.load(f)
            self.global_mean, self.global_std, ok = torch.tensor(stats["mean"]), torch.tensor(stats["std"]), True
        except (OSError, KeyError, ValueError):
            self.global_mean, self.global_std = torch.zeros(3), torch.ones(3)
        self.is_fitted = True
        return ok

    def __call__(self, points, mode="norm"):
        if not self.is_fitted:
            self.load_stats()
        mean, std = self.global_mean.to(points.device), self.global_std.to(points.device)
        return (points - mean) / std if mode == "norm" else points * std + mean


def save_point_clouds(points, prefix, directory="."):
    """One `<prefix>_<i>.npy` file of float32 [N, 3] per generated sample (README.md:108-113). Returns the paths."""
    pts = points.detach().float().cpu().numpy() if torch.is_tensor(points) else np.asarray(points, dtype="float32")
    if pts.ndim != 3 or pts.shape[-1] != 3:
        raise ValueError(f"expected [B, N, 3] points, got {pts.shape}")
    os.makedirs(directory, exist_ok=True)
    paths = []
    for i, pc in enumerate(pts):
        paths.append(os.path.join(directory, f"{prefix}_{i}.npy"))
        np.save(paths[-1], pc)
    return paths
