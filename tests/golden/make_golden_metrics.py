"""Golden vectors for the point-set metrics, made by RUNNING THE REFERENCE'S OWN FUNCTIONS (build container only):

    python tests/golden/make_golden_metrics.py

Imports /root/reference/test_optimize.py as-is (torch, numpy, scipy, tqdm and matplotlib are installed; the module has
only definitions at import time) and records compute_chamfer_distance / compute_emd_distance (test_optimize.py:354-415)
on seeded point sets: equal and unequal point counts, coordinates beyond the +-5 clamp, coincident points.
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_optimize as R  # noqa: E402  (the reference's evaluation script)

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    g = torch.Generator().manual_seed(77)
    cases = {}
    specs = [("equal", 3, 96, 96, 1.0), ("ragged", 2, 130, 77, 1.0), ("clamped", 2, 64, 64, 4.0), ("large", 1, 700, 700, 0.6)]
    for name, B, n, m, scale in specs:
        pred = torch.randn(B, n, 3, generator=g) * scale
        target = torch.randn(B, m, 3, generator=g) * scale
        if name == "equal":
            target[0, :10] = pred[0, :10]  # coincident points: zero distances through the 1 / (d + 1e-6) weights
        cases[name] = (pred, target)
    arrays = {}
    for name, (pred, target) in cases.items():
        arrays[f"{name}/pred"], arrays[f"{name}/target"] = pred.numpy(), target.numpy()
        arrays[f"{name}/chamfer"] = R.compute_chamfer_distance(pred, target).double().numpy()
        arrays[f"{name}/emd"] = R.compute_emd_distance(pred, target).double().numpy()
        print(name, tuple(pred.shape), tuple(target.shape), float(arrays[f"{name}/chamfer"]), float(arrays[f"{name}/emd"]))
    # the normaliser (test_optimize.py:32-75) on explicit statistics
    norm = R.GlobalNormalizer()
    norm.global_mean, norm.global_std, norm.is_fitted = torch.tensor([0.1, -0.2, 0.3]), torch.tensor([0.5, 2.0, 1.5]), True
    pts = cases["equal"][0]
    arrays["norm/mean"], arrays["norm/std"] = norm.global_mean.numpy(), norm.global_std.numpy()
    arrays["norm/out"] = norm(pts, "norm").numpy()
    arrays["norm/back"] = norm(norm(pts, "norm"), "denorm").numpy()
    path = os.path.join(HERE, "pointset_metrics.npz")
    np.savez(path, **arrays)
    print("->", path, f"{os.path.getsize(path) / 1e3:.1f} kB")


if __name__ == "__main__":
    main()
