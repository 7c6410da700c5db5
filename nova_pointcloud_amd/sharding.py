"""Batch sharding of the generation path over the GPUs of a node.

Samples are independent (no cross-sample op anywhere in the path), so the path shards by batch row:
rank r generates a contiguous block of the prompts with replicated weights, and the only exchange
is one all-gather of the generated point sets ([B_loc, N, 3] f32 per rank) — RCCL over xGMI on
GPUs (`backend="nccl"`), gloo in the CPU tests.

Seed contract: every rank seeds its generator identically and passes `batch_shard=(lo, hi, total)`; the generation
order uniforms (embeddings.py:265) and the per-step noise (transformer_3d.py:131) are then drawn for the GLOBAL batch
and sliced, so `sharded(seed) == unsharded(seed)` sample for sample (`generate_sharded`).
"""
import torch


def shard_range(total, rank, world):
    """Contiguous block [lo, hi) of `total` items owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_list(items, rank, world):
    lo, hi = shard_range(len(items), rank, world)
    return list(items[lo:hi])


def gather_points(points, group=None, total=None, per=1):
    """All-gather [B_loc, N, 3] point sets of every rank into [sum B_loc, N, 3] (rank order).

    `total` = number of items (prompts) sharded over the ranks, `per` samples each (`num_images_per_prompt`): every rank's
    shard size then follows from `shard_range` and the path's
    one collective is issued without a size exchange or a host read-back (nothing between the pipeline's last kernel and the
    all-gather waits for the device). Equal shards use one `all_gather_into_tensor`; ragged shards are padded to the
    largest. Without `total` the shards must be equal-sized (checked against the gathered tensor's shape only).
    """
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():  # with a process group the exchange runs, also for one rank
        return points
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    points = points.contiguous()
    if total is None:
        counts = [points.shape[0]] * world
    else:
        counts = [(hi - lo) * per for lo, hi in (shard_range(total, r, world) for r in range(world))]
        if counts[rank] != points.shape[0]:
            raise ValueError(f"rank {rank} holds {points.shape[0]} samples, shard_range({total}, {rank}, {world}) says {counts[rank]}")
    if len(set(counts)) == 1:
        out = points.new_empty((world * counts[0],) + tuple(points.shape[1:]))
        dist.all_gather_into_tensor(out, points, group=group)
        return out
    pad = points.new_zeros((max(counts),) + tuple(points.shape[1:]))
    pad[: points.shape[0]] = points
    out = pad.new_empty((world * max(counts),) + tuple(points.shape[1:]))
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = out.view(world, max(counts), *points.shape[1:])
    return torch.cat([parts[r, :n] for r, n in enumerate(counts)])


def generate_sharded(pipe, prompt_embeds, rank, world, group=None, **call_kwargs):
    """Run `pipe` on this rank's contiguous block of `prompt_embeds` (the GLOBAL list) and return the point sets of
    the whole batch [len(prompt_embeds), N, 3] on every rank. `call_kwargs` are `NOVAPipeline.__call__` arguments; a
    `generator` must be seeded identically on all ranks (its draws cover the global batch)."""
    from diffnext.pipelines.nova.pipeline_nova import points_from_latents

    per = int(call_kwargs.get("num_images_per_prompt", 1))
    lo, hi = shard_range(len(prompt_embeds), rank, world)
    out = pipe(prompt_embeds=list(prompt_embeds[lo:hi]), output_type="latent", disable_progress_bar=True,
               batch_shard=(lo * per, hi * per, len(prompt_embeds) * per), **call_kwargs)
    return gather_points(points_from_latents(out.frames).float().contiguous(), group, total=len(prompt_embeds), per=per)
