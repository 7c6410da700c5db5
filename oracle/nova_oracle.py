"""ORACLE — test infrastructure, NOT product code.

A CPU restatement, in plain functional PyTorch (float32 or float64), of the arithmetic of the
reference's generation hot path (zailaiyiwan123/NOVA_pointcloud, package `diffnext`). Every
function cites the reference file:line it follows (paths relative to the reference root).

Who may use this file: tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg, as the
checker / the timed CPU baseline. The product (nova_pointcloud_amd/) never imports it.

Parity pin: this restatement is checked against golden vectors produced in the build container
by running the reference's OWN modules (tests/golden/make_golden.py imports
diffnext.models.{vision_transformer,diffusion_mlp,embeddings,normalization,guidance_scaler} and
diffnext.models.transformers.transformer_3d from /root/reference; they import without diffusers).
The reference's schedulers and NOVAPipeline import `diffusers`, which is absent here, so
`cosine_schedule`, `cfm_sigmas` and `encode_prompt_embeds` are restated from the source text and
are "parity unpinned by execution" (pinned only by hand-derived values in tests/test_oracle.py).

All tensors live on the CPU. Parameters come as a flat dict with the reference's state_dict keys.
"""

import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# host-side schedule / sampler arithmetic
# ----------------------------------------------------------------------------------------------
def cosine_schedule(num_patches, num_inference_steps):
    """diffnext/pipelines/nova/pipeline_nova.py:129-132 — tokens predicted at each AR step."""
    k = num_inference_steps
    mask_ratios = np.cos(0.5 * np.pi * np.arange(k + 1) / k)
    mask_length = np.round(mask_ratios * num_patches).astype("int64")
    return mask_length[:-1] - mask_length[1:]


def cfm_sigmas(num_steps, shift=1.0, num_train_timesteps=1000):
    """diffnext/schedulers/scheduling_cfm.py:40-49,92-104 — returns (timesteps f32[S], sigmas list[S+1])."""
    train_t = np.arange(1, num_train_timesteps + 1, dtype="float32")[::-1]
    train_sig = train_t / num_train_timesteps
    train_sig = shift * train_sig / (1 + (shift - 1) * train_sig)  # __init__ uses the ctor shift
    sigma_min, sigma_max = float(train_sig[-1]), float(train_sig[0])
    t_max, t_min = sigma_max * num_train_timesteps, sigma_min * num_train_timesteps
    timesteps = np.linspace(t_max, t_min, num_steps, dtype="float32")
    sigmas = timesteps / num_train_timesteps
    sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
    return sigmas * num_train_timesteps, sigmas.tolist() + [0]


def ddpm_plan(num_steps, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
              prediction_type="epsilon", variance_type="fixed_small", timestep_spacing="leading", steps_offset=0):
    """diffnext/schedulers/scheduling_ddpm.py:134-157 (betas), :182-209 (set_timesteps), :211-234 (_get_variance),
    :236-316 (step). Returns per step (t, kx, kv, c0, cx, sigma) with x0 = kx x + kv v, x <- c0 x0 + cx x + sigma eps.
    The reference's step has no clipping / thresholding stage. PARITY UNPINNED BY EXECUTION (class imports diffusers)."""
    n = num_train_timesteps
    if beta_schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    elif beta_schedule == "scaled_linear":
        betas = torch.linspace(beta_start**0.5, beta_end**0.5, n, dtype=torch.float32) ** 2
    else:
        raise NotImplementedError(beta_schedule)
    acp = torch.cumprod(1.0 - betas, dim=0)
    if timestep_spacing == "leading":
        ts = (np.arange(0, num_steps) * (n // num_steps)).round()[::-1].copy().astype(np.int64) + steps_offset
    elif timestep_spacing == "linspace":
        ts = np.linspace(0, n - 1, num_steps).round()[::-1].copy().astype(np.int64)
    elif timestep_spacing == "trailing":
        ts = np.arange(n, 0, -n / num_steps).round().astype(np.int64) - 1
    else:
        raise ValueError(timestep_spacing)
    plan = []
    for t in [int(v) for v in ts]:
        prev_t = t - n // num_steps
        a_t, a_prev = acp[t], (acp[prev_t] if prev_t >= 0 else torch.tensor(1.0))
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1 - cur_alpha
        if prediction_type == "epsilon":
            kx, kv = 1 / a_t**0.5, -(b_t**0.5) / a_t**0.5
        elif prediction_type == "sample":
            kx, kv = torch.tensor(0.0), torch.tensor(1.0)
        else:  # v_prediction
            kx, kv = a_t**0.5, -(b_t**0.5)
        var = torch.clamp(b_prev / b_t * cur_beta, min=1e-20)
        sigma = (cur_beta if variance_type == "fixed_large" else var) ** 0.5 if t > 0 else torch.tensor(0.0)
        plan.append((t, float(kx), float(kv), float(a_prev**0.5 * cur_beta / b_t), float(cur_alpha**0.5 * b_prev / b_t), float(sigma)))
    return plan


def encode_prompt_embeds(text_weight, prompt_embeds, num_tokens):
    """pipeline_nova.py:204-215 + embeddings.py:179-188 for the `prompt_embeds`, guidance > 1 path.

    Returns [2B, num_tokens, token_dim]: rows 0..B-1 = padded prompts, B..2B-1 = unconditional.
    """
    x = text_weight[:num_tokens].expand(len(prompt_embeds), -1, -1).clone()
    for i, p in enumerate(prompt_embeds):
        x[i, : p.shape[0]] = torch.as_tensor(p).to(x.dtype)
    neg = text_weight[: x.shape[1]].expand(x.shape[0], -1, -1)
    return torch.cat([x, neg])


# ----------------------------------------------------------------------------------------------
# layers
# ----------------------------------------------------------------------------------------------
def layer_norm(x, w=None, b=None, eps=1e-5):
    return F.layer_norm(x, x.shape[-1:], w, b, eps)


def rope_weight(pos, head_dim, pad=0, ids=None, theta=10000.0):
    """embeddings.py:45-67 RotaryEmbed3D.get_func — rotation table [bs, 1, pad+n, hd/2, 2, 2]."""
    pos = pos.gather(1, ids.expand(-1, -1, 3)) if ids is not None else pos
    pos = F.pad(pos, (0, 0, pad, 0), value=0) if pad else pos
    weight = []
    dims = [head_dim // 8] + [(head_dim - head_dim // 8) // 2] * 2
    for i, grid in enumerate(pos.split(1, dim=-1)):
        scale = torch.arange(0, dims[i], 2).float().div_(dims[i])  # float32 buffer in the reference
        freq = torch.pow(theta, scale.float())
        freq = grid * freq.reciprocal().unsqueeze(0)
        freq = torch.stack([freq.cos(), -freq.sin(), freq.sin(), freq.cos()], dim=-1)
        weight += [freq.view(freq.shape[:-1] + (2, 2))]
    return torch.cat(weight, dim=-3).unsqueeze(1)


def rope_pos(t, bs, hw):
    """embeddings.py:52-57 RotaryEmbed3D.get_pos — (t, h, w) integer grid [bs, t*h*w, 3] float32."""
    thw = [t] + list(hw)
    pos = torch.zeros(thw + [3])
    grid = [torch.arange(n) for n in thw]
    for i in range(3):
        pos[..., i].add_(grid[i].view([-1 if i == j else 1 for j in range(3)]))
    return pos.view(1, -1, 3).expand(bs, -1, -1)


def apply_rope(x, weight):
    """embeddings.py:36-43 ApplyFunc — x [S,h,L,d]; adjacent-pair rotation."""
    x = x.view(*x.shape[:-1], -1, 1, 2)
    w = weight.to(dtype=x.dtype)
    return w[..., 0].mul(x[..., 0]).add(w[..., 1] * x[..., 1]).flatten(3)


def attention(p, pre, x, heads, rope=None, cache=None):
    """vision_transformer.py:51-64 Attention.forward (no mask). `cache`: None, or a list that holds [k, v] of the
    earlier calls (:55-60: the new k, v are concatenated behind them along the token axis)."""
    S, L, D = x.shape
    qkv = F.linear(x, p[pre + "qkv.weight"], p[pre + "qkv.bias"])
    q, k, v = qkv.view(S, L, 3, heads, D // heads).permute(2, 0, 3, 1, 4).unbind(0)
    if rope is not None:
        q, k = apply_rope(q, rope), apply_rope(k, rope)
    if cache is not None:
        if cache:
            k, v = torch.cat([cache[0], k], dim=2), torch.cat([cache[1], v], dim=2)
        cache[:] = [k, v]
    o = F.scaled_dot_product_attention(q, k, v)
    return F.linear(o.transpose(1, 2).flatten(2), p[pre + "proj.weight"], p[pre + "proj.bias"])


def vit_block(p, pre, x, heads, rope=None, cache=None):
    """vision_transformer.py:78-92 Block.forward — POST-norm: x = LN(f(x)) + x."""
    a = attention(p, pre + "attn.", x, heads, rope, cache)
    x = layer_norm(a, p[pre + "norm1.weight"], p[pre + "norm1.bias"]) + x
    m = F.linear(F.gelu(F.linear(x, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"])),
                 p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])
    return layer_norm(m, p[pre + "norm2.weight"], p[pre + "norm2.bias"]) + x


def vit_forward(p, pre, depth, heads, x, c=None, prev_ids=None, pos=None, pos_embed=None, caches=None):
    """vision_transformer.py:128-146 VisionTransformer.forward for token input x [S,N,D].

    The MAE-style split: blocks[:depth/2] see [c ; known tokens], blocks[depth/2:] see
    [c' ; full canvas]. pos_embed: abs-PE table [N, D] added in place of nn.Identity (:131).
    """
    head_dim = x.shape[-1] // heads
    x = x + pos_embed if pos_embed is not None else x
    x_masked = x
    pe1 = pe2 = None
    if pos is not None:  # prepare_pe :119-123
        pad = 0 if c is None else c.size(1)
        pe1 = pe2 = rope_weight(pos, head_dim, pad)
        if prev_ids is not None:
            pe1 = rope_weight(pos, head_dim, pad, prev_ids)
    if prev_ids is not None:
        x = x.gather(1, prev_ids.expand(-1, -1, x.size(-1)))
    x = x if c is None else torch.cat([c, x], dim=1)
    enc = depth // 2
    for i in range(enc):
        x = vit_block(p, f"{pre}blocks.{i}.", x, heads, pe1, None if caches is None else caches[i])
    if prev_ids is not None and c is not None:
        c, x = x.split((c.size(1), x.size(1) - c.size(1)), dim=1)
    if prev_ids is not None:
        x = x_masked.scatter(1, prev_ids.expand(-1, -1, x.size(-1)), x)
        x = x if c is None else torch.cat([c, x], dim=1)
    for i in range(enc, depth):
        x = vit_block(p, f"{pre}blocks.{i}.", x, heads, pe2, None if caches is None else caches[i])
    x = x if c is None else x[:, c.size(1):]
    return layer_norm(x, p[pre + "norm.weight"], p[pre + "norm.bias"])


def projector(p, pre, x):
    """diffusion_mlp.py:26-36 Projector: fc2(SiLU(fc1(x)))."""
    return F.linear(F.silu(F.linear(x, p[pre + "fc1.weight"], p[pre + "fc1.bias"])), p[pre + "fc2.weight"], p[pre + "fc2.bias"])


def time_cond_embed(p, pre, timestep, z, freq_dim=256):
    """diffusion_mlp.py:56-75 TimeCondEmbed."""
    dim = freq_dim // 2
    freq = torch.arange(dim, dtype=torch.float32).mul(-9.210340371976184 / dim).exp().unsqueeze(0)
    emb = timestep.unsqueeze(-1).float() * freq
    emb = torch.cat([emb.cos(), emb.sin()], dim=-1).to(dtype=z.dtype)
    t = projector(p, pre + "timestep_proj.", emb)
    return projector(p, pre + "condition_proj.", z) + (t.unsqueeze(1) if t.dim() == 2 else t)


def adaln_zero(p, pre, x, z, num_stats, eps=1e-6):
    """normalization.py:24-36 AdaLayerNormZero (no LoRA): returns (modulated x, remaining stats)."""
    stats = F.linear(F.silu(z), p[pre + "proj.weight"], p[pre + "proj.bias"]).chunk(num_stats, dim=-1)
    return layer_norm(x, None, None, eps) * (1 + stats[0]) + stats[1], stats[2:]


def patch_embed(p, pre, x, patch):
    """embeddings.py:160-166 PatchEmbed.forward on [B,C,H,W] -> [B,N,D] (conv k = s = patch)."""
    return F.conv2d(x, p[pre + "proj.weight"], p[pre + "proj.bias"], stride=patch).flatten(2).transpose(1, 2)


def patchify(x, patch):
    """embeddings.py:152-154: [B,C,H,W] -> [B,N,p*p*C]."""
    B, C, H, W = x.shape
    x = x.view(B, C, H // patch, patch, W // patch, patch)
    return x.permute(0, 2, 4, 3, 5, 1).flatten(1, 2).flatten(2, 4).contiguous()


def unpatchify(x, patch, C, h, w):
    """embeddings.py:156-158: [B,N,p*p*C] -> [B,C,H,W]."""
    x = x.view(-1, h, w, patch, patch, C)
    return x.permute(0, 5, 1, 3, 2, 4).flatten(2, 3).flatten(3, 4).contiguous()


def diffusion_mlp(p, pre, depth, x_img, timestep, z, pred_ids, patch):
    """diffusion_mlp.py:89-99 DiffusionMLP.forward with pred_ids: returns [S,N,p*p*C]."""
    x = patch_embed(p, pre + "patch_embed.", x_img, patch)
    o = patchify(x_img, patch)
    x = x.gather(1, pred_ids.expand(-1, -1, x.size(-1)))
    z = z.gather(1, pred_ids.expand(-1, -1, z.size(-1)))
    z = time_cond_embed(p, pre + "time_cond_embed.", timestep, z)
    for i in range(depth):
        b = f"{pre}blocks.{i}."
        h, (gate,) = adaln_zero(p, b + "norm1.", x, z, 3)
        h = projector(p, b + "proj.", h)
        x = layer_norm(h, p[b + "norm2.weight"], p[b + "norm2.bias"]) * gate + x
    x = adaln_zero(p, pre + "norm.", x, z, 2)[0]
    x = F.linear(x, p[pre + "head.weight"], p[pre + "head.bias"])
    return o.scatter(1, pred_ids.expand(-1, -1, x.size(-1)), x)


def sincos_2d(dim, h, w):
    """embeddings.py:70-88 PosEmbed.get_space_embed at base size (h, w): [h*w, dim] float32."""
    freq_hw = 1 / (10000 ** (torch.arange(dim // 4, dtype=torch.float32) / (dim // 4)))
    grid_h = torch.arange(h, dtype=torch.float32)
    grid_w = torch.arange(w, dtype=torch.float32)
    grid_w, grid_h = torch.meshgrid(grid_w, grid_h, indexing="xy")
    freq_w, freq_h = [g.reshape(-1, 1) * freq_hw.unsqueeze(0) for g in (grid_w, grid_h)]
    return torch.cat([freq_w.sin(), freq_w.cos(), freq_h.sin(), freq_h.cos()], dim=-1)


def video_time_embed(p, pre, t, base_t):
    """embeddings.py:103-111 VideoPosEmbed.get_time_embed: [t, 1, D]."""
    freq_t = 1 / (10000 ** (torch.arange(128, dtype=torch.float32).unsqueeze(0) / 128))
    grid = torch.arange(t, dtype=torch.float32) / (t / base_t)
    f = grid.view(-1, 1, 1).mul(freq_t)
    sincos = torch.cat([f.sin(), f.cos()], dim=-1).to(p[pre + "norm.weight"].dtype)
    h = F.linear(F.silu(F.linear(sincos, p[pre + "time_proj.0.weight"], p[pre + "time_proj.0.bias"])),
                 p[pre + "time_proj.2.weight"], p[pre + "time_proj.2.bias"])
    return layer_norm(h, p[pre + "norm.weight"], p[pre + "norm.bias"])


def motion_embed(p, pre, rows, flow=None, fps=None, base_flow=5, base_fps=12):
    """embeddings.py:119-137 MotionEmbed.forward: two tokens per row, [rows, 2, D] (flow token then fps token)."""
    freq_m = 1 / (10000 ** (torch.arange(128, dtype=torch.float32).unsqueeze(0) / 128))
    out = []
    for k, vals, base in (("flow", flow, base_flow), ("fps", fps, base_fps)):
        vals = [base] * rows if vals is None else vals
        f = torch.as_tensor(vals).view(-1, 1, 1).float().mul(freq_m)
        sincos = torch.cat([f.sin(), f.cos()], dim=-1).to(p[f"{pre}{k}_proj.0.weight"].dtype)
        out.append(F.linear(F.silu(F.linear(sincos, p[f"{pre}{k}_proj.0.weight"], p[f"{pre}{k}_proj.0.bias"])),
                            p[f"{pre}{k}_proj.2.weight"], p[f"{pre}{k}_proj.2.bias"]))
    return torch.cat(out, dim=1)


def frame_mixer(p, pre, x, z):
    """normalization.py:33-36,41-46 AdaLayerNorm(dim, rank, eps=None).forward(x, z) as assembled at transformer_nova.py:87-89:
    no normalisation, x * (1 + scale) + shift with (scale, shift) = proj(lora(SiLU(z)))."""
    h = F.silu(z)
    if pre + "lora.weight" in p:
        h = F.linear(h, p[pre + "lora.weight"])
    scale, shift = F.linear(h, p[pre + "proj.weight"], p[pre + "proj.bias"]).chunk(2, dim=-1)
    return x * (1 + scale) + shift


# ----------------------------------------------------------------------------------------------
# the generation loop
# ----------------------------------------------------------------------------------------------
class Config(dict):
    """Geometry of a model instance (what NOVATransformer3DModel.__init__ derives, transformer_nova.py:59-102)."""

    __getattr__ = dict.__getitem__


def make_config(image_dim, latent_hw, patch, embed_dim, heads, video_depth, image_depth, decoder_depth,
                text_token_len, rotary=True, video_base_t=1):
    H, W = latent_hw
    return Config(image_dim=image_dim, latent_hw=(H, W), patch=patch, video_patch=patch * 2, embed_dim=embed_dim,
                  heads=heads, video_depth=video_depth, image_depth=image_depth, decoder_depth=decoder_depth,
                  text_token_len=text_token_len, rotary=rotary, video_base_t=video_base_t,
                  image_hw=(H // patch, W // patch), video_hw=(H // (2 * patch), W // (2 * patch)))


def generate(p, cfg, prompt, num_preds, num_diffusion_steps=25, guidance_scale=5.0, generator=None,
             shift=1.0, dtype=torch.float32, u_dist=None, noises=None, trace=None, guidance_trunc=0, guidance_renorm=1,
             ddpm=None, max_latent_length=1, image_guidance_scale=0, spatiotemporal_guidance_scale=0, motion_flow=None,
             fps=None, latents=None, c_pre=None):
    """Transformer3DModel.forward in eval mode (transformer_3d.py:63-77,102-164,192-200).

    c_pre: the caller's own condition list inputs["c"] (transformer_3d.py:66: model-width rows [S, Lc_i, D]; the text embedding of
    `prompt` is appended behind them, :70-71; `prompt` None = no text rows, as when the model has no text embedding).

    prompt: [2B, Lt, token_dim] from encode_prompt_embeds ([B, ...] when guidance_scale <= 1). Returns x [B, C, T, H, W]
    with T = max_latent_length (a point set is T = 1).
    RNG contract: one uniform_ [B,N,1] (embeddings.py:265) then one normal_ [B,C,H,W] per AR step of every generated
    frame (transformer_3d.py:131), all from `generator`; `u_dist` / `noises` replay pre-drawn values instead (used to
    compare dtypes / devices on identical noise). `trace` (dict) collects intermediates for the tests.
    T > 1: KV-cached conditioning encoder (vision_transformer.py:55-60), frame positions (transformer_3d.py:143-150),
    frame mixer when the weights hold `video_encoder.mixer.*` (:156-158), `latents` = [first frame] given (:159-160).
    motion_flow (list, one value per sample): MotionEmbed tokens appended to the text prefix when the weights hold
    `motion_embed.*` (:72-75). image / spatiotemporal guidance: the 3-pass forms of guidance_scaler.py:37-57,78-85.
    """
    p = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in p.items()}
    D, heads, C, patch = cfg.embed_dim, cfg.heads, cfg.image_dim, cfg.patch
    (H, W), (h, w), (hv, wv) = cfg.latent_hw, cfg.image_hw, cfg.video_hw
    N, Nv, T = h * w, hv * wv, max_latent_length
    cfg_on = guidance_scale > 1
    extra_pass = image_guidance_scale + spatiotemporal_guidance_scale > 0
    passes = (3 if extra_pass else 2) if cfg_on else 1
    rows0 = prompt.shape[0] if prompt is not None else c_pre[0].shape[0]
    B = rows0 // 2 if cfg_on else rows0
    expand = (lambda t: torch.cat([t] * passes)) if cfg_on else (lambda t: t)

    # preprocess :63-77 — the given list, then TextEmbed.forward embeddings.py:203-206 (+ MotionEmbed :72-75), torch.cat(dim=1) :77
    parts = [t.to(dtype) for t in (c_pre or [])]
    if prompt is not None:
        parts.append(layer_norm(F.linear(prompt.to(dtype), p["text_embed.proj.weight"], p["text_embed.proj.bias"]),
                                p["text_embed.norm.weight"], p["text_embed.norm.bias"]))
    c_txt = torch.cat(parts, dim=1) if len(parts) > 1 else parts[0]
    if motion_flow is not None and "motion_embed.flow_proj.0.weight" in p:
        flow = list(motion_flow) * (2 if cfg_on else 1)
        fps_rows = list(fps) * (2 if cfg_on else 1) if fps else None
        c_txt = torch.cat([c_txt, motion_embed(p, "motion_embed.", c_txt.shape[0], flow, fps_rows).to(dtype)], dim=1)
    if passes == 3:  # expand_text guidance_scaler.py:46-51: [cond, uncond] + [uncond] (image) or + [cond] (spatiotemporal)
        cond_rows, uncond_rows = c_txt.chunk(2)
        c_txt = torch.cat([cond_rows, uncond_rows, uncond_rows if image_guidance_scale else cond_rows])
    S = c_txt.shape[0]
    timesteps, sigmas = cfm_sigmas(num_diffusion_steps, shift)
    ddpm_steps = ddpm_plan(num_diffusion_steps, **ddpm) if ddpm is not None else None
    if ddpm_steps is not None:
        timesteps = [np.int64(st[0]) for st in ddpm_steps]

    # generate_video :135-164
    bos, mask_token = p["mask_embed.bos_token"], p["mask_embed.mask_token"]
    img_pe = None if cfg.rotary else sincos_2d(D, h, w).to(dtype)
    time_pos = rope_pos(T, 1, cfg.video_hw).chunk(T, 1) if cfg.rotary else None
    time_embed = None if cfg.rotary else video_time_embed(p, "video_pos_embed.", T, cfg.video_base_t)
    caches = [[] for _ in range(cfg.video_depth)] if T > 1 else None
    mixing = "video_encoder.mixer.proj.weight" in p
    x = torch.zeros(B, C, H, W, dtype=dtype)
    noise = torch.empty(B, C, H, W, dtype=dtype)
    pos = rope_pos(1, S, cfg.image_hw) if cfg.rotary else None
    order = None
    latents = list(latents) if latents else []
    given = bool(latents)
    c_first = None
    step_no = 0
    if trace is not None:
        trace["z"], trace["c_frames"] = [], []
    for t in range(T):
        cv = patch_embed(p, "video_encoder.patch_embed.", x, cfg.video_patch)  # :151 (patch 2p conv of the previous frame)
        if t == 0:
            cv = bos.expand(B, Nv, D).clone()  # :152 overwritten by bos_token
        if not cfg.rotary:
            cv = cv + time_embed[t]  # :153 add_(time_embed[t])
            cv = cv + sincos_2d(D, hv, wv).to(dtype)  # VideoPosEmbed.forward embeddings.py:113-115
        if cfg_on:  # expand(c, padding=bos) guidance_scaler.py:39-44
            cv = torch.stack([cv] * passes)
            if image_guidance_scale:
                cv[1] = bos
            cv = cv.flatten(0, 1)
        c = vit_forward(p, "video_encoder.", cfg.video_depth, heads, cv, None if t else c_txt, None,
                        time_pos[t] if cfg.rotary else None, caches=caches)
        if trace is not None:
            trace["c"] = c.clone()
            trace["c_frames"].append(c.clone())
        if mixing:  # :156-158
            c_first = c if t == 0 else c_first
            c = frame_mixer(p, "video_encoder.mixer.", c_first, c) if t else c
        if t == 0 and given:  # :159-160
            x = latents[-1].to(dtype).clone()
            continue

        # generate_frame :115-133
        x = torch.zeros(B, C, H, W, dtype=dtype)
        mask = torch.ones(B, N, 1, dtype=dtype)
        if order is None:
            if u_dist is None:
                u_dist = torch.empty_like(mask).uniform_(generator=generator)
            order = u_dist.argsort(dim=1)  # embeddings.py:265-266
            if trace is not None:
                trace["order"] = order.clone()
        pred_pos, prev_ids = 0, None
        sched = [int(v) for v in num_preds if v > 0]
        for i, n in enumerate(sched):
            g_now = guidance_scale  # decay_guidance_scale with min_guidance_scale = None: constant (:31-35)
            z = patch_embed(p, "image_encoder.patch_embed.", x, patch)
            z = z * (1 - mask) + mask_token * mask  # embeddings.py:272-274 with the mask BEFORE this step's update
            pred_ids = order[:, pred_pos:pred_pos + n]
            pred_mask = torch.zeros_like(mask).scatter_(1, pred_ids, 1)
            pred_pos, mask = pred_pos + n, mask * (1 - pred_mask)
            pred_ids = expand(pred_ids)
            prev_ids = prev_ids if i else pred_ids.new_empty((pred_ids.size(0), 0, 1))
            z = vit_forward(p, "image_encoder.", cfg.image_depth, heads, expand(z), c, prev_ids, pos, img_pe)
            if trace is not None:
                trace["z"].append(z.clone())
            prev_ids = torch.cat([prev_ids, pred_ids], dim=1)
            if noises is None:
                noise.normal_(generator=generator)
            else:
                noise = noises[step_no].to(dtype)
            step_no += 1
            # denoise :102-113 (guidance passes; guidance_trunc / guidance_renorm per guidance_scaler.py:59-72)
            xt = noise
            zz, ids, cfg_live = z, pred_ids, cfg_on
            for j, tt in enumerate(timesteps):
                if cfg_live and guidance_trunc and float(tt) < guidance_trunc:  # maybe_disable :59-65: stays off afterwards
                    cfg_live, zz, ids = False, zz.chunk(passes)[0], ids.chunk(passes)[0]
                timestep = torch.as_tensor(tt).expand(zz.shape[0])
                x_in = expand(xt) if cfg_live else xt
                pred = diffusion_mlp(p, "image_decoder.", cfg.decoder_depth, x_in, timestep, zz, ids, patch)
                if cfg_live:  # scale :74-87
                    def renorm(v, cond):  # :67-72, norms over every row of the sample (echo rows included)
                        if guidance_renorm >= 1:
                            return v
                        dims = tuple(range(1, v.dim()))
                        ratio = cond.norm(dim=dims, keepdim=True) / v.norm(dim=dims, keepdim=True)
                        return v * ratio.clamp(guidance_renorm, 1)

                    if image_guidance_scale:  # :78-81
                        cond, uncond, imgcond = pred.chunk(3)
                        pred = renorm(uncond + (cond - imgcond) * g_now, cond) + (imgcond - uncond) * image_guidance_scale
                    elif spatiotemporal_guidance_scale:  # :82-85
                        cond, uncond, perturb = pred.chunk(3)
                        pred = renorm(uncond + (cond - uncond) * g_now, cond) + (cond - perturb) * spatiotemporal_guidance_scale
                    else:  # :86-87
                        cond, uncond = pred.chunk(2)
                        pred = renorm(uncond + (cond - uncond) * g_now, cond)
                pred = unpatchify(pred, patch, C, h, w)
                if ddpm_steps is None:
                    dt = sigmas[j + 1] - sigmas[j]  # scheduling_cfm.py:134-135
                    xt = pred * dt + xt
                else:  # scheduling_ddpm.py:268-312
                    _, kx, kv, c0, cx, sigma = ddpm_steps[j]
                    x0 = kx * xt + kv * pred
                    xt = c0 * x0 + cx * xt
                    if int(tt) > 0:
                        xt = xt + sigma * torch.randn(xt.shape, generator=generator, dtype=dtype)
            sample = patchify(xt, patch)
            x = x + unpatchify(sample * pred_mask, patch, C, h, w)  # :133
        latents.append(x.clone())
    return torch.stack(latents, dim=2)
