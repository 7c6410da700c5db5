"""CPU: the FLOP accounting bench.py reports against (SURVEY §8d / BASELINE.md §3 table) and its CLI contract."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nova_pointcloud_amd"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

from oracle import nova_oracle as O  # noqa: E402


@pytest.mark.parametrize("D,N,K,S,tflop", [(768, 256, 4, 4, 1.097), (768, 1024, 64, 25, 69.88), (1024, 2048, 64, 25, 275.0),
                                           (1536, 2048, 64, 25, 565.3)])
def test_flops_per_sample_matches_baseline_table(D, N, K, S, tflop):
    sched = [int(v) for v in O.cosine_schedule(N, K) if v > 0]
    got = bench.flops_per_sample(D, N, N // 4, 256, sched, S) / 1e12
    assert abs(got - tflop) / tflop < 2e-3, got


def test_workloads_name_the_baseline_configs():
    assert bench.WORKLOADS["d48w1024_2048pts_b32"] == (1024, 16, 32, 64, 32)   # BASELINE.json configs[2] (the metric's config)
    assert bench.WORKLOADS["d48w768_1024pts_b8"] == (768, 12, 32, 32, 8)       # configs[1]
    assert bench.MFMA_BF16_PEAK_TFLOPS == 2500.0


def test_host_cores_is_bounded():
    assert 1 <= bench.host_cores() <= 32
