"""Autoregressive generator base class (reference diffnext/models/transformers/transformer_3d.py).

`Transformer3DModel` keeps the reference's constructor, attributes (`video_encoder`,
`image_encoder`, `image_decoder`, `mask_embed`, `text_embed`, `label_embed`, `video_pos_embed`,
`image_pos_embed`, `motion_embed`, `noise_scheduler`, `sample_scheduler`, `pipeline_preprocess`,
`loss_repeat`) and methods (`preprocess` :63-77, `get_losses` :79-100, `denoise` :102-113,
`generate_frame` :115-133, `generate_video` :135-164, `train_video` :166-190, `forward` :192-200).

Eval-mode `forward` on an MI355X hands the whole set-by-set generation loop to
`nova_pointcloud_amd.engine.NovaEngine` (hand-written gfx950 kernels, no eager fallback);
everything else (CPU tensors, training) runs the PyTorch definitions below.
"""
from typing import Dict

import torch
from torch import nn
from torch.nn import functional as F

from ... import _backend
from ..guidance_scaler import GuidanceScaler

try:
    from tqdm import tqdm
except ImportError:  # pragma: no cover
    tqdm = None


class Transformer3DModel(nn.Module):
    def __init__(self, video_encoder=None, image_encoder=None, image_decoder=None, mask_embed=None, text_embed=None,
                 label_embed=None, video_pos_embed=None, image_pos_embed=None, motion_embed=None,
                 noise_scheduler=None, sample_scheduler=None):
        super().__init__()
        self.video_encoder = video_encoder
        self.image_encoder = image_encoder
        self.image_decoder = image_decoder
        self.mask_embed = mask_embed
        self.text_embed = text_embed
        self.label_embed = label_embed
        self.video_pos_embed = video_pos_embed
        self.image_pos_embed = image_pos_embed
        self.motion_embed = motion_embed
        self.noise_scheduler = noise_scheduler
        self.sample_scheduler = sample_scheduler
        self.pipeline_preprocess = lambda inputs: inputs
        self.loss_repeat = 4

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def dtype(self) -> torch.dtype:
        return next(p for p in self.parameters() if p.is_floating_point()).dtype

    def progress_bar(self, iterable, enable=True):
        return tqdm(iterable) if (enable and tqdm is not None) else iterable

    # ------------------------------------------------------------------ inputs
    def preprocess(self, inputs: Dict):
        """Allocate the latent canvas and turn prompts / motion values into the condition prefix `c`."""
        add_guidance = inputs.get("guidance_scale", 1) > 1
        conds = inputs.get("c", [])
        if inputs.get("x", None) is None:
            shape = (inputs.get("batch_size", 1), self.image_encoder.image_dim) + tuple(self.image_encoder.image_size)
            inputs["x"] = torch.empty(shape, device=self.device, dtype=self.dtype)
        if inputs.get("prompt", None) is not None and self.text_embed:
            conds.append(self.text_embed(inputs.pop("prompt")))
        if inputs.get("motion_flow", None) is not None and self.motion_embed:
            flow, fps = inputs.pop("motion_flow", None), inputs.pop("fps", None)
            flow, fps = [v + v if (add_guidance and v) else v for v in (flow, fps)]
            conds.append(self.motion_embed(conds[-1], flow, fps))
        inputs["c"] = torch.cat(conds, dim=1) if len(conds) > 1 else conds[0]

    # ------------------------------------------------------------------ training
    def get_losses(self, z: torch.Tensor, x: torch.Tensor, video_shape=None) -> Dict:
        rep = lambda t: t.repeat(self.loss_repeat, *((1,) * (t.dim() - 1)))
        pe = self.image_encoder.patch_embed
        z, x = rep(z), pe.patchify(rep(x))
        noise = torch.randn(x.shape, dtype=x.dtype, device=x.device)
        timestep = self.noise_scheduler.sample_timesteps(z.shape[:2], device=z.device)
        x_t = pe.unpatchify(self.noise_scheduler.add_noise(x, noise, timestep))
        timestep = getattr(self.noise_scheduler, "timestep", timestep)
        pred_type = getattr(self.noise_scheduler.config, "prediction_type", "flow")
        pred = self.image_decoder(x_t, timestep, z)
        target = noise.float() if pred_type == "epsilon" else noise.sub(x).float()
        loss = F.mse_loss(pred.float(), target, reduction="none").mean(-1, True)
        weight = rep(self.mask_embed.mask.to(loss.dtype))
        loss = loss * weight / (weight.sum() + 1e-5)
        if video_shape is not None:
            per_frame = loss.view((-1,) + video_shape).transpose(0, 1).sum((1, 2))
            i2i = per_frame[1:].sum() * (video_shape[0] / (video_shape[0] - 1))
            return {"loss_t2i": per_frame[0] * video_shape[0], "loss_i2i": i2i}
        return {"loss": loss.sum()}

    def train_video(self, inputs):
        x = inputs["x"] = inputs["x"].unsqueeze(2) if inputs["x"].dim() == 4 else inputs["x"]
        bs, T = x.size(0), x.size(2)
        # temporal autoregression: frames 0..T-2 (+ begin-of-video tokens) condition frames 0..T-1
        c = self.video_encoder.patch_embed(x[:, :, : T - 1])
        bov = self.mask_embed.bos_token.expand(bs, 1, c.size(-2), -1)
        c, pos = self.video_pos_embed(torch.cat([bov, c], dim=1)), None
        if self.image_pos_embed:
            pos = self.video_pos_embed.get_pos(c.size(1), bs, self.video_encoder.patch_embed.hw)
        attn_mask = self.mask_embed.get_attn_mask(c, inputs["c"]) if T > 1 else None
        for blk in self.video_encoder.blocks:
            blk.attn.attn_mask = attn_mask
        c = self.video_encoder(c.flatten(1, 2), inputs["c"], pos=pos)
        if not isinstance(self.video_encoder.mixer, nn.Identity) and T > 1:
            first, rest = c.view(bs, T, -1, c.size(-1)).split([1, T - 1], 1)
            c = torch.cat([first, self.video_encoder.mixer(first, rest)], 1)
        # masked autoregression inside each frame
        frames = x[:, :, :T].transpose(1, 2).flatten(0, 1)
        z, bs = self.image_encoder.patch_embed(frames), bs * T
        if self.image_pos_embed:
            pos = self.image_pos_embed.get_pos(1, bs, self.image_encoder.patch_embed.hw)
        z = self.image_encoder(self.mask_embed(z), c.reshape(bs, -1, c.size(-1)), pos=pos)
        return self.get_losses(z, frames, video_shape=(T, z.size(1)) if T > 1 else None)

    # ------------------------------------------------------------------ generation (PyTorch definition)
    @torch.no_grad()
    def denoise(self, z, x, guidance_scaler, generator=None, pred_ids=None) -> torch.Tensor:
        """Diffusion sampling of the tokens in `pred_ids`, starting from noise canvas x [B,C,H,W]."""
        pe = self.image_encoder.patch_embed
        self.sample_scheduler._step_index = None
        for t in self.sample_scheduler.timesteps:
            z, pred_ids = guidance_scaler.maybe_disable(t, z, pred_ids)
            timestep = torch.as_tensor(t, device=x.device).expand(z.shape[0])
            pred = self.image_decoder(guidance_scaler.expand(x), timestep, z, pred_ids)
            pred = pe.unpatchify(guidance_scaler.scale(pred))
            x = self.sample_scheduler.step(pred, t, x, generator=generator).prev_sample
        return pe.patchify(x)

    @torch.inference_mode()
    def generate_frame(self, states: Dict, inputs: Dict):
        """Masked set-by-set generation of one frame (for point sets: the whole sample)."""
        scaler = GuidanceScaler(**inputs)
        generator = self.mask_embed.generator = inputs.get("generator", None)
        shard = self.mask_embed.batch_shard = inputs.get("batch_shard", None)
        if shard is not None and type(self.sample_scheduler).__name__ == "DDPMScheduler":
            raise NotImplementedError("batch_shard with the ancestral sampler is built on the HIP path only")
        schedule = [n for n in inputs["num_preds"] if n > 0]
        pe = self.image_encoder.patch_embed
        c, x, self.mask_embed.mask = states["c"], states["x"].zero_(), None
        pos = self.image_pos_embed.get_pos(1, c.size(0)) if self.image_pos_embed else None
        prev_ids = None
        for i, n in enumerate(self.progress_bar(schedule, inputs.get("tqdm2", False))):
            scaler.decay_guidance_scale((i + 1) / len(schedule))
            z = self.mask_embed(pe(x))
            pred_mask, pred_ids = self.mask_embed.get_pred_mask(n)
            pred_ids = scaler.expand(pred_ids)
            if prev_ids is None:
                prev_ids = pred_ids.new_empty((pred_ids.size(0), 0, 1))
            z = self.image_encoder(scaler.expand(z), c, prev_ids, pos=pos)
            prev_ids = torch.cat([prev_ids, pred_ids], dim=1)
            if shard is None:
                states["noise"].normal_(generator=generator)
            else:  # global-batch draw, this shard's rows (sharded(seed) == unsharded(seed))
                full = states["noise"].new_empty((shard[2],) + tuple(states["noise"].shape[1:])).normal_(generator=generator)
                states["noise"].copy_(full[shard[0]:shard[1]])
            sample = self.denoise(z, states["noise"], scaler.clone(), generator, pred_ids)
            x.add_(pe.unpatchify(sample * pred_mask))

    @torch.inference_mode()
    def generate_video(self, inputs: Dict):
        """Frame-by-frame generation; a point set is the max_latent_length == 1 case."""
        scaler = GuidanceScaler(**inputs)
        T = inputs.get("max_latent_length", 1)
        self.sample_scheduler.set_timesteps(inputs.get("num_diffusion_steps", 25))
        states = {"x": inputs["x"], "noise": inputs["x"].clone()}
        latents, self.mask_embed.pred_ids = inputs.get("latents", []), None
        rope = bool(self.image_pos_embed)
        time_pos = self.video_pos_embed.get_pos(T).chunk(T, 1) if rope else None
        time_embed = None if rope else self.video_pos_embed.get_time_embed(T)
        inputs["c"] = scaler.expand_text(inputs["c"])
        self.video_encoder.enable_kvcache(T > 1)
        for t in self.progress_bar(range(T), inputs.get("tqdm1", True)):
            states["t"] = t
            pos = time_pos[t] if rope else None
            c = self.video_encoder.patch_embed(states["x"])
            if t == 0:
                c[:] = self.mask_embed.bos_token
            if not rope:
                c = self.video_pos_embed(c.add_(time_embed[t]))
            c = scaler.expand(c, padding=self.mask_embed.bos_token)
            c = states["c"] = self.video_encoder(c, None if t else inputs["c"], pos=pos)
            if not isinstance(self.video_encoder.mixer, nn.Identity):
                states["c"] = self.video_encoder.mixer(states["*"], c) if t else c
                states["*"] = states["*"] if t else states["c"]
            if t == 0 and latents:
                states["x"].copy_(latents[-1])
            else:
                self.generate_frame(states, inputs)
                latents.append(states["x"].clone())
        self.video_encoder.enable_kvcache(False)

    # ------------------------------------------------------------------ entry point
    def forward(self, inputs):
        self.pipeline_preprocess(inputs)
        if not self.training and self.device.type == "cuda":
            inputs["latents"] = inputs.pop("latents", [])
            return {"x": _backend.engine().NovaEngine.for_model(self).generate(inputs)}
        self.preprocess(inputs)
        if self.training:
            return self.train_video(inputs)
        inputs["latents"] = inputs.pop("latents", [])
        self.generate_video(inputs)
        return {"x": torch.stack(inputs["latents"], dim=2)}
