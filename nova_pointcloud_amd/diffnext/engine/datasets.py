"""Point-set training data sources with the `.next() -> [inputs]` interface the Trainer consumes (the reference's
loader, codewithgpu records behind config.train_dataloader, is a third-party package that is absent here).

`inputs` is the dict `Transformer3DModel.forward` takes in training: {"x": [B, 3, H, W] latent canvas (one xyz point per
token), "prompt": list of B prompt embeddings [len_b, token_dim]}.
"""
import glob
import os

import numpy as np
import torch


class SyntheticPointClouds(object):
    """Random shapes (points on noisy spheres / boxes) with random prompt embeddings: for smoke runs and benchmarks."""

    def __init__(self, batch_size, latent_hw, token_dim, seed=0, device="cpu", dtype=torch.float32, shard_id=0, num_shards=1,
                 max_prompt_len=8):
        self.batch_size, self.hw, self.token_dim, self.max_len = batch_size, tuple(latent_hw), token_dim, max_prompt_len
        self.gen = torch.Generator().manual_seed(seed * 1000003 + shard_id)
        self.device, self.dtype = device, dtype

    def next(self):
        B, (H, W) = self.batch_size, self.hw
        pts = torch.randn(B, H * W, 3, generator=self.gen)
        pts = pts / pts.norm(dim=-1, keepdim=True).clamp_min(1e-6) * (0.5 + 0.5 * torch.rand(B, 1, 1, generator=self.gen))
        x = pts.transpose(1, 2).reshape(B, 3, H, W).to(self.device, self.dtype)
        lens = torch.randint(1, self.max_len + 1, (B,), generator=self.gen).tolist()
        prompt = [(0.02 * torch.randn(n, self.token_dim, generator=self.gen)).to(self.device, self.dtype) for n in lens]
        return [{"x": x, "prompt": prompt}]


class NpyPointClouds(object):
    """Directory of `<name>.npy` point clouds [n >= H*W, 3] (+ optional `<name>.prompt.npy` embeddings [len, token_dim]),
    sharded over ranks by index; points are subsampled to the H*W tokens of the latent canvas."""

    def __init__(self, root, batch_size, latent_hw, token_dim, seed=0, device="cpu", dtype=torch.float32, shard_id=0, num_shards=1,
                 max_prompt_len=None):
        files = sorted(f for f in glob.glob(os.path.join(root, "*.npy")) if not f.endswith(".prompt.npy"))
        if not files:
            raise ValueError("Unsupported dataset: " + root)
        self.files = files[shard_id::num_shards] or files
        self.batch_size, self.hw, self.token_dim = batch_size, tuple(latent_hw), token_dim
        self.rng = np.random.RandomState(seed + shard_id)
        self.device, self.dtype, self.pos, self.max_len = device, dtype, 0, max_prompt_len

    def next(self):
        H, W = self.hw
        xs, prompts = [], []
        for _ in range(self.batch_size):
            f = self.files[self.pos % len(self.files)]
            self.pos += 1
            pts = np.load(f).astype("float32")
            idx = self.rng.choice(len(pts), H * W, replace=len(pts) < H * W)
            xs.append(torch.from_numpy(pts[idx]).t().reshape(3, H, W))
            pf = f[:-4] + ".prompt.npy"
            emb = np.load(pf).astype("float32") if os.path.exists(pf) else np.zeros((1, self.token_dim), "float32")
            prompts.append(torch.from_numpy(emb[: self.max_len]).to(self.device, self.dtype))
        return [{"x": torch.stack(xs).to(self.device, self.dtype), "prompt": prompts}]
