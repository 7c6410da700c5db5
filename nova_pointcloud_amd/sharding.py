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


def gather_points(points, group=None):
    """All-gather [B_loc, N, 3] point sets of every rank into [sum B_loc, N, 3] (rank order).

    Equal shard sizes use one `all_gather_into_tensor`; ragged shards are padded to the largest.
    """
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():  # with a process group the exchange runs, also for one rank
        return points
    world = dist.get_world_size(group)
    points = points.contiguous()
    sizes = torch.tensor([points.shape[0]], dtype=torch.int64, device=points.device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    counts = [int(s.item()) for s in all_sizes]
    if len(set(counts)) == 1:
        out = points.new_empty((world * counts[0],) + tuple(points.shape[1:]))
        dist.all_gather_into_tensor(out, points, group=group)
        return out
    pad = points.new_zeros((max(counts),) + tuple(points.shape[1:]))
    pad[: points.shape[0]] = points
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, counts)])


def generate_sharded(pipe, prompt_embeds, rank, world, group=None, **call_kwargs):
    """Run `pipe` on this rank's contiguous block of `prompt_embeds` (the GLOBAL list) and return the point sets of
    the whole batch [len(prompt_embeds), N, 3] on every rank. `call_kwargs` are `NOVAPipeline.__call__` arguments; a
    `generator` must be seeded identically on all ranks (its draws cover the global batch)."""
    from diffnext.pipelines.nova.pipeline_nova import points_from_latents

    per = int(call_kwargs.get("num_images_per_prompt", 1))
    lo, hi = shard_range(len(prompt_embeds), rank, world)
    out = pipe(prompt_embeds=list(prompt_embeds[lo:hi]), output_type="latent", disable_progress_bar=True,
               batch_shard=(lo * per, hi * per, len(prompt_embeds) * per), **call_kwargs)
    return gather_points(points_from_latents(out.frames).float().contiguous(), group)
