"""NOVA generation pipeline (reference diffnext/pipelines/nova/pipeline_nova.py:27-239).

`NOVAPipeline.__init__` and `__call__` keep the reference's keyword sets and defaults. The call
builds the cosine set-size schedule (:129-132), pads / stacks the prompt embeddings with the
unconditional rows (:175-220) and hands one `inputs` dict to `self.transformer` (:139) — which,
on an MI355X, runs the whole autoregressive loop on libnova_hip.so. `output_type="latent"`
returns the generated tensor itself ([B, C, 1, H, W]; for point sets C = 3 and each of the H*W
tokens is one xyz point), skipping the VAE exactly like the reference (:140-141).
"""
from typing import List

import numpy as np
import torch

from ..._compat import DiffusionPipeline
from .pipeline_utils import NOVAPipelineOutput, PipelineMixin


def cosine_set_sizes(num_patches, num_inference_steps):
    """Tokens revealed at each AR step: differences of round(N * cos(pi/2 * i/K)), i = 0..K."""
    ratios = np.cos(0.5 * np.pi * np.arange(num_inference_steps + 1) / num_inference_steps)
    remaining = np.round(ratios * num_patches).astype("int64")
    return remaining[:-1] - remaining[1:]


def points_from_latents(frames):
    """[B, 3, 1, H, W] latent output -> [B, N, 3] point coordinates (SURVEY §8d mapping)."""
    return frames[:, :, 0].flatten(2).transpose(1, 2)


class NOVAPipeline(DiffusionPipeline, PipelineMixin):
    _optional_components = ["transformer", "scheduler", "vae", "text_encoder", "tokenizer"]
    model_cpu_offload_seq = "text_encoder->transformer->vae"

    def __init__(self, transformer=None, scheduler=None, vae=None, text_encoder=None, tokenizer=None,
                 trust_remote_code=True):
        super().__init__()
        self.vae = self.register_module(vae, "vae")
        self.text_encoder = self.register_module(text_encoder, "text_encoder")
        self.tokenizer = self.register_module(tokenizer, "tokenizer")
        self.transformer = self.register_module(transformer, "transformer")
        self.scheduler = self.register_module(scheduler, "scheduler")
        self.transformer.sample_scheduler, self.guidance_scale = self.scheduler, 5.0
        if self.transformer.text_embed:
            self.tokenizer_max_length = self.transformer.text_embed.num_tokens
            self.transformer.text_embed.encoders = [self.tokenizer, self.text_encoder]
        self.image_processor = None  # VAE decode / PIL conversion are image features (not on the point-set path)

    @torch.no_grad()
    def __call__(self, prompt=None, num_inference_steps=64, num_diffusion_steps=25, max_latent_length=1,
                 guidance_scale=5, guidance_trunc=0, guidance_renorm=1, image_guidance_scale=0,
                 spatiotemporal_guidance_scale=0, flow_shift=None, motion_flow=5, negative_prompt=None, image=None,
                 num_images_per_prompt=1, generator=None, latents=None, prompt_embeds=None,
                 negative_prompt_embeds=None, disable_progress_bar=False, output_type="pil", **kwargs
                 ) -> NOVAPipelineOutput:
        """Generate one sample per prompt. See the reference docstring (:79-125) for the arguments."""
        self.guidance_scale = guidance_scale
        if flow_shift:
            self.scheduler.set_shift(flow_shift)
        inputs = dict(kwargs)
        inputs.update(
            generator=generator, num_inference_steps=num_inference_steps, num_diffusion_steps=num_diffusion_steps,
            max_latent_length=max_latent_length, guidance_scale=guidance_scale, guidance_trunc=guidance_trunc,
            guidance_renorm=guidance_renorm, image_guidance_scale=image_guidance_scale,
            spatiotemporal_guidance_scale=spatiotemporal_guidance_scale, flow_shift=flow_shift, image=image,
            num_images_per_prompt=num_images_per_prompt, disable_progress_bar=disable_progress_bar,
            output_type=output_type)
        num_patches = int(np.prod(self.transformer.config.image_base_size))
        inputs["num_preds"] = cosine_set_sizes(num_patches, num_inference_steps)
        inputs["tqdm1"] = max_latent_length > 1 and not disable_progress_bar
        inputs["tqdm2"] = max_latent_length == 1 and not disable_progress_bar
        inputs["prompt"] = self.encode_prompt(prompt, num_images_per_prompt, negative_prompt, prompt_embeds,
                                              negative_prompt_embeds)
        inputs["latents"] = self.prepare_latents(image, num_images_per_prompt, generator, latents)
        inputs["batch_size"] = len(inputs["prompt"]) // (2 if guidance_scale > 1 else 1)
        inputs["motion_flow"] = [motion_flow] * inputs["batch_size"]
        x = self.transformer(inputs)["x"]
        if output_type != "latent":
            x = self._decode(x, output_type)
        name = {4: "images", 5: "frames"}[x.dim() if torch.is_tensor(x) else np.ndim(x)]
        return NOVAPipelineOutput(**{name: x})

    def _decode(self, x, output_type):
        """VAE decode + PIL/uint8 conversion (reference :140-143, image_processor.py) are image features that
        sit outside the point-set hot path: only tensor outputs are built here."""
        if output_type == "pt":
            return x
        raise NotImplementedError(
            f'output_type="{output_type}" needs the VAE / image processor, which this build does not ship; '
            'point sets are returned with output_type="latent" (frames [B, 3, 1, H, W])')

    def prepare_latents(self, image=None, num_images_per_prompt=1, generator=None, latents=None) -> List[torch.Tensor]:
        if latents is not None:
            return latents
        return [] if image is None else [self.encode_image(image, num_images_per_prompt, generator)]

    def encode_prompt(self, prompt, num_images_per_prompt=1, negative_prompt=None, prompt_embeds=None,
                      negative_prompt_embeds=None) -> torch.Tensor:
        """[cond rows ; unconditional rows] padded to the text token count (2B rows when guidance > 1)."""
        embedder, cfg_on = self.transformer.text_embed, self.guidance_scale > 1
        if prompt_embeds is not None:
            cond = embedder.encode_prompts(prompt_embeds)
            if not cfg_on:
                # the reference leaves `c` unbound here (pipeline_nova.py:213-215); the evident intent is the cond rows
                return cond.repeat_interleave(num_images_per_prompt, dim=0)
            if negative_prompt_embeds is not None:
                neg = embedder.encode_prompts(negative_prompt_embeds)
            else:
                neg = embedder.weight[: cond.shape[1]].expand(cond.shape[0], -1, -1)
            return torch.cat([cond, neg]).repeat_interleave(num_images_per_prompt, dim=0)
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        if cfg_on:
            neg = negative_prompt or ""
            prompts = prompts + ([neg] * len(prompts) if isinstance(neg, str) else list(neg))
        return embedder.encode_prompts(prompts).repeat_interleave(num_images_per_prompt, dim=0)

    def encode_image(self, image, num_images_per_prompt=1, generator=None) -> torch.Tensor:
        x = torch.as_tensor(image, device=self.device).to(dtype=self.dtype)
        x = x.sub(127.5).div_(127.5).permute(2, 0, 1).unsqueeze_(0)
        x = self.vae.scale_(self.vae.encode(x).latent_dist.sample(generator))
        return x.expand(num_images_per_prompt, -1, -1, -1)
