"""Point-set metrics: the oracle against outputs of the reference's own functions (CPU), and the
HIP kernels against the oracle and the same fixture (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pointset_metrics.npz")
CASES = ["equal", "ragged", "clamped", "large"]


@pytest.fixture(scope="module")
def gold():
    z = np.load(GOLD, allow_pickle=False)
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_functions(gold, name):
    pred, target = gold[f"{name}/pred"], gold[f"{name}/target"]
    cd, emd = MO.compute_chamfer_distance(pred, target), MO.compute_emd_distance(pred, target)
    assert abs(float(cd) - float(gold[f"{name}/chamfer"])) <= 1e-6 * float(gold[f"{name}/chamfer"])
    assert abs(float(emd) - float(gold[f"{name}/emd"])) <= 1e-6 * float(gold[f"{name}/emd"])


def test_normalizer_matches_reference(gold):
    from nova_pointcloud_amd.metrics import GlobalNormalizer

    n = GlobalNormalizer()
    n.global_mean, n.global_std, n.is_fitted = gold["norm/mean"], gold["norm/std"], True
    pts = gold["equal/pred"]
    assert torch.equal(n(pts, "norm"), gold["norm/out"])
    assert torch.equal(n(n(pts, "norm"), "denorm"), gold["norm/back"])
    m = GlobalNormalizer()
    assert m.load_stats("/nonexistent/stats.json") is False and torch.equal(m(pts), pts)  # identity statistics


def test_npy_export_round_trip(tmp_path):
    from nova_pointcloud_amd.metrics import save_point_clouds

    pts = torch.randn(3, 40, 3)
    paths = save_point_clouds(pts, "shape", str(tmp_path))
    assert [os.path.basename(p) for p in paths] == ["shape_0.npy", "shape_1.npy", "shape_2.npy"]
    for i, p in enumerate(paths):
        back = np.load(p)
        assert back.dtype == np.float32 and back.shape == (40, 3) and np.array_equal(back, pts[i].numpy())
    with pytest.raises(ValueError):
        save_point_clouds(torch.zeros(4, 3), "bad", str(tmp_path))


def test_metrics_refuse_cpu_tensors():
    from nova_pointcloud_amd import hip, metrics

    with pytest.raises(hip.NovaHipError):
        metrics.compute_chamfer_distance(torch.zeros(1, 8, 3), torch.zeros(1, 8, 3))


# --------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_metrics_match_reference_outputs(gold, hip, name):
    from nova_pointcloud_amd import metrics

    pred, target = gold[f"{name}/pred"].cuda(), gold[f"{name}/target"].cuda()
    cd, emd = metrics.compute_chamfer_distance(pred, target), metrics.compute_emd_distance(pred, target)
    assert cd.is_cuda and abs(float(cd) - float(gold[f"{name}/chamfer"])) <= 1e-5 * float(gold[f"{name}/chamfer"])
    assert abs(float(emd) - float(gold[f"{name}/emd"])) <= 1e-5 * float(gold[f"{name}/emd"])


@pytest.mark.gpu
def test_hip_distance_kernels_match_exact_differences(hip):
    """nn_dist / pairwise_dist against float64 explicit differences: ragged sizes across the 256-point block and the
    1024-point LDS tile, clamping, unit normalisation, one point, 2048 x 2048 (the metric's size)."""
    from nova_pointcloud_amd import metrics

    g = torch.Generator().manual_seed(5)
    for B, N, M, clamp, unit in [(2, 1, 1, 5.0, False), (3, 257, 1025, 5.0, False), (1, 300, 40, 1.0, True), (2, 2048, 2048, 2.0, False)]:
        x, y = torch.randn(B, N, 3, generator=g) * 1.5, torch.randn(B, M, 3, generator=g) * 1.5
        d = metrics.nn_dist(x.cuda(), y.cuda(), clamp, unit).cpu().double()
        ref = MO.nn_dist(x, y, clamp, unit)
        assert (d - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-7, (B, N, M)
        if not unit:
            D = metrics.pairwise_dist(x.cuda(), y.cuda(), clamp).cpu().double()
            refD = (x.double().clamp(-clamp, clamp)[:, :, None] - y.double().clamp(-clamp, clamp)[:, None]).norm(dim=-1)
            assert D.shape == (B, N, M) and (D - refD).abs().max() <= 1e-5 * refD.abs().max()
            assert torch.allclose(D.min(dim=2)[0], d, rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_hip_train_loss_metrics_match_oracle_and_properties(hip):
    """distChamfer / emd_approx (train_newloss.py:316-372; oracle restated, parity unpinned by execution) + properties the
    domain offers at the metric's size: a set against itself is at the floor, EMD of a permuted set is 0, symmetry."""
    from nova_pointcloud_amd import metrics

    g = torch.Generator().manual_seed(9)
    a, b = torch.randn(2, 256, 3, generator=g) * 0.4, torch.randn(2, 256, 3, generator=g) * 0.4
    dl, dr = metrics.distChamfer(a.cuda(), b.cuda())
    rl, rr = MO.dist_chamfer(a, b)
    assert abs(float(dl) - float(rl)) <= 1e-4 * float(rl) and abs(float(dr) - float(rr)) <= 1e-4 * float(rr)
    e = metrics.emd_approx(a.cuda(), b.cuda()).cpu()
    assert torch.allclose(e, MO.emd_approx(a, b).float(), rtol=1e-5)
    assert abs(float(metrics.robust_chamfer_distance(a.cuda(), b.cuda())) - float((rl + rr) / 2)) <= 1e-4 * float((rl + rr) / 2)
    big = torch.randn(1, 2048, 3, generator=g).cuda()
    perm = big[:, torch.randperm(2048, generator=g).cuda()]
    assert float(metrics.emd_approx(big, perm).max()) <= 1.1e-8  # distances are floored at 1e-8 (train_newloss.py:364)
    s1, s2 = metrics.distChamfer(big * 0.3, perm * 0.3)
    floor = float(torch.exp(torch.tensor(-10.0)))  # log(d + 1e-8) is clamped at -10 (train_newloss.py:343): e^-10 is the floor
    assert abs(float(s1) - floor) <= 1e-6 * floor and abs(float(s2) - floor) <= 1e-6 * floor
    x, y = metrics.distChamfer(a.cuda(), b.cuda()), metrics.distChamfer(b.cuda(), a.cuda())
    assert float(x[0]) == float(y[1]) and float(x[1]) == float(y[0])
