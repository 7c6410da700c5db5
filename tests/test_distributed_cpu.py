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

from nova_pointcloud_amd.sharding import shard_range  # noqa: E402


def free_port():
    """A TCP port nobody listens on right now (rendezvous of the spawned ranks on 127.0.0.1)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_range_covers_everything():
    for total in (0, 1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _pipe(gold):
    from diffnext.pipelines import NOVAPipeline
    from diffnext.schedulers import FlowMatchEulerDiscreteScheduler
    from test_mirror_cpu import build_from_golden

    return NOVAPipeline(transformer=build_from_golden(gold), scheduler=FlowMatchEulerDiscreteScheduler())


def _worker(rank, world, port, ragged, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from golden_util import Golden
    from nova_pointcloud_amd.sharding import generate_sharded

    gold = Golden("tiny_rope")
    m = gold.meta
    prompts = gold.prompt_embeds + ([gold.prompt_embeds[0][:3]] if ragged else [])
    allp = generate_sharded(_pipe(gold), prompts, rank, world, num_inference_steps=m["K"], num_diffusion_steps=m["S"],
                            guidance_scale=m["guidance"], generator=torch.Generator().manual_seed(100))
    if rank == 0:
        ret.put(allp)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ragged", [False, True])
def test_two_rank_sharded_generation_equals_unsharded_run_of_the_same_seed(ragged):
    """the order uniforms and the per-step noise are drawn for the GLOBAL batch and sliced per rank,
    so a batch sharded over 2 ranks (even or ragged shards) reproduces the single-process run of the same seed."""
    from golden_util import Golden
    from nova_pointcloud_amd.sharding import generate_sharded

    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ragged, ret)) for r in range(2)]
    [p.start() for p in procs]
    got = ret.get(timeout=300)
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    gold = Golden("tiny_rope")
    m = gold.meta
    prompts = gold.prompt_embeds + ([gold.prompt_embeds[0][:3]] if ragged else [])
    want = generate_sharded(_pipe(gold), prompts, 0, 1, num_inference_steps=m["K"], num_diffusion_steps=m["S"],
                            guidance_scale=m["guidance"], generator=torch.Generator().manual_seed(100))
    assert got.shape == want.shape == (len(prompts), m["latent_h"] * m["latent_w"], 3)
    # bitwise on the draws; the arithmetic runs at another batch size per rank (CPU GEMM blocking), hence a tolerance
    assert (got - want).abs().max() <= 1e-5 * want.abs().max()
    if not ragged:  # and the unsharded seeded run is the reference's golden run when the seed is the fixture's
        ref = generate_sharded(_pipe(gold), gold.prompt_embeds, 0, 1, num_inference_steps=m["K"], num_diffusion_steps=m["S"],
                               guidance_scale=m["guidance"], generator=torch.Generator().manual_seed(m["sample_seed"]))
        gx = gold.t["out/x"][:, :, 0].flatten(2).transpose(1, 2)
        assert (ref - gx).abs().max() <= 1e-4 * gx.abs().max()


def _gather_worker(rank, world, port, total, per, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nova_pointcloud_amd.sharding import gather_points

    lo, hi = shard_range(total, rank, world)
    mine = torch.arange(lo * per, hi * per, dtype=torch.float32).view(-1, 1, 1).expand(-1, 5, 3).contiguous()
    out = gather_points(mine, total=total, per=per)
    bad = None
    try:  # a rank whose row count contradicts shard_range is an error, not a silent mis-assembly
        gather_points(mine[:-1] if mine.shape[0] else mine.new_zeros(1, 5, 3), total=total, per=per)
    except ValueError as e:
        bad = str(e)
    if rank == 0:
        ret.put((out, bad))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total,per", [(3, 7, 2), (4, 8, 1), (3, 2, 1)])
def test_gather_points_sizes_come_from_shard_range(world, total, per):
    """The path's only collective (the reference's rank-strided precedent: evaluations/geneval/sample.py:55,70)
    is issued without a size exchange: every rank derives all shard sizes from `shard_range` (ragged shards, several samples
    per prompt, a rank with no prompt at all), pads to the largest and gathers into one tensor."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, total, per, ret)) for r in range(world)]
    [p.start() for p in procs]
    out, bad = ret.get(timeout=120)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert out.shape == (total * per, 5, 3)
    assert torch.equal(out[:, 0, 0], torch.arange(total * per, dtype=torch.float32))
    assert bad is not None and "shard_range" in bad


@pytest.mark.parametrize("world,global_batch", [(2, 0), (2, 3), (4, 6), (8, 11)])
def test_bench_launch_contract_dry_run(world, global_batch):
    """The driver's N > 1 launch line, rehearsed on CPU (gloo) with 2, 4 and 8 ranks: rendezvous, per-rank shard, gather,
    max-over-ranks timing; the JSON line carries what the process group itself saw (`ranks_seen`), each rank's block of the global
    prompt list (`shards`: [rank, lo, hi), incl. ragged global batches: 3 over 2, 6 over 4, 11 over 8 ranks), and rank 0's check
    that the gathered rows are the unsharded run's samples in prompt order."""
    import json
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1",
           "--cpu-dry-run", "--workload", "d48w768_256pts_b1", "--ar-steps", "2", "--diffusion-steps", "2",
           "--global-batch", str(global_batch)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # exactly one JSON line, from rank 0
    rec = json.loads(lines[0])
    G = global_batch or world
    assert rec["n_gpus"] == world and rec["points"] == [G, 256, 3]
    assert rec["ranks_seen"] == list(range(world))
    assert rec["shards"] == [[r, *shard_range(G, r, world)] for r in range(world)]
    assert [hi - lo for _, lo, hi in rec["shards"]].count(0) == 0 and sum(hi - lo for _, lo, hi in rec["shards"]) == G
    assert rec["sharded_vs_unsharded_max_rel_diff"] < 1e-4
