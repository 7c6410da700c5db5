"""CPU, world_size 2 (gloo): batch sharding + all-gather of generated point sets (the N > 1 path of bench.py)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nova_pointcloud_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from nova_pointcloud_amd.sharding import gather_points, shard_list, shard_range  # noqa: E402


def test_shard_range_covers_everything():
    for total in (0, 1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _run_pipe(gold, prompts, seed):
    from diffnext.pipelines import NOVAPipeline
    from diffnext.pipelines.nova.pipeline_nova import points_from_latents
    from diffnext.schedulers import FlowMatchEulerDiscreteScheduler
    from test_mirror_cpu import build_from_golden

    m = gold.meta
    pipe = NOVAPipeline(transformer=build_from_golden(gold), scheduler=FlowMatchEulerDiscreteScheduler())
    out = pipe(prompt_embeds=prompts, num_inference_steps=m["K"], num_diffusion_steps=m["S"], guidance_scale=m["guidance"],
               generator=torch.Generator().manual_seed(seed), output_type="latent", disable_progress_bar=True)
    return points_from_latents(out.frames).float().contiguous()


def _worker(rank, world, port, ragged, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from golden_util import Golden

    gold = Golden("tiny_rope")
    prompts = gold.prompt_embeds + ([gold.prompt_embeds[0][:3]] if ragged else [])
    mine = shard_list(prompts, rank, world)
    pts = _run_pipe(gold, mine, seed=100 + rank)
    allp = gather_points(pts)
    if rank == 0:
        ret.put(allp)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ragged", [False, True])
def test_two_rank_sharded_generation_matches_per_shard_runs(ragged):
    from golden_util import Golden

    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (7 if ragged else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ragged, ret)) for r in range(2)]
    [p.start() for p in procs]
    got = ret.get(timeout=300)
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    gold = Golden("tiny_rope")
    prompts = gold.prompt_embeds + ([gold.prompt_embeds[0][:3]] if ragged else [])
    want = torch.cat([_run_pipe(gold, shard_list(prompts, r, 2), seed=100 + r) for r in range(2)])
    assert got.shape == want.shape == (len(prompts), gold.meta["latent_h"] * gold.meta["latent_w"], 3)
    assert torch.equal(got, want)


def test_bench_launch_contract_two_ranks_dry_run():
    """The driver's N > 1 launch line, rehearsed on CPU (gloo): rendezvous, per-rank shard, gather, max-over-ranks timing."""
    import json
    import subprocess

    port = 29700 + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--cpu-dry-run", "--workload", "d48w768_256pts_b1", "--ar-steps", "2", "--diffusion-steps", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # exactly one JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["points"] == [2, 256, 3]
