"""Data-parallel gradient exchange for one-process-per-GPU training (RCCL over xGMI through torch.distributed's "nccl"
backend on MI355X, gloo on CPU): the role `accelerator.prepare` / DDP play in the reference's Trainer
(diffnext/engine/train_engine.py:47,110-130).

Gradients are packed into a few large flat buckets (xGMI is point to point, ring all-reduce is per-link bound: few large
messages, MI355X_MICROARCH-style sizing) and averaged with one all_reduce per bucket after the last micro-step of an
iteration; parameters are broadcast from rank 0 once at construction. No hooks, no graph rewriting: the trainer calls
`sync_gradients()` where the reference calls `accelerator.backward` on the accumulation boundary.
"""
import torch
import torch.distributed as dist


class GradientReducer(object):
    def __init__(self, params, bucket_mb=256.0, group=None):
        self.group = group
        # collectives run whenever a process group exists - also with one rank (a single-GPU torchrun launch then still
        # goes through RCCL: broadcast, bucketed all_reduce, scalar all_reduce), never without one
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, cur_bytes = [], [], 0
        limit = int(bucket_mb * 2 ** 20)
        for p in self.params:
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > limit or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        if self.active:
            with torch.no_grad():
                for p in params:  # every rank starts from rank 0's weights (frozen parameters included)
                    dist.broadcast(p.data, src=0, group=group)

    @torch.no_grad()
    def sync_gradients(self):
        """Average the accumulated gradients over the ranks, bucket by bucket. A parameter without a gradient contributes zeros
        to the exchange (the payload must have one shape on every rank) and keeps `grad = None` afterwards unless some rank
        did produce one - so parameters off the current path (frame mixer, frame patch-embed at T = 1) get no weight decay or
        moment updates, exactly as in a single-process run and in the reference."""
        if not self.active or not self.params:
            return
        # every bucket's all_reduce is queued first; the has-gradient flags of ALL parameters travel as one extra small tensor and are
        # read back once, after the last collective is queued (a host read per bucket would serialise bucket k + 1's concat and
        # collective behind bucket k's completion)
        dev = self.params[0].device
        flags = torch.tensor([0.0 if p.grad is None else 1.0 for p in self.params], dtype=torch.float32, device=dev)
        flats = []
        for bucket in self.buckets:
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flats.append(flat)
        dist.all_reduce(flags, op=dist.ReduceOp.SUM, group=self.group)
        used_by = dict(zip((id(p) for p in self.params), (flags > 0).tolist()))
        for bucket, flat in zip(self.buckets, flats):
            flat.div_(self.world)
            off = 0
            for p in bucket:
                g = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                if p.grad is None:
                    if used_by[id(p)]:
                        p.grad = g.clone()
                else:
                    p.grad.copy_(g)

    def gather_mean(self, value):
        """Mean of a scalar over the ranks (the `accelerator.gather(v).mean()` of run_model)."""
        if not self.active:
            return float(value)
        t = torch.as_tensor(float(value), dtype=torch.float64, device=self.params[0].device if self.params else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t) / self.world
