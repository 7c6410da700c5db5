"""Golden vectors for the multi-frame (max_latent_length > 1) and 3-pass-guidance branches of the generation path,
made by RUNNING THE REFERENCE'S OWN MODULES (build container only; needs /root/reference, read-only):

    python tests/golden/make_golden_video.py

Runs, imported as-is from the reference: Transformer3DModel.forward / generate_video / generate_frame / denoise
(transformer_3d.py:63-77,102-164,192-200) with the KV-cached conditioning encoder (vision_transformer.py:55-60,125-126),
the AdaLayerNorm frame mixer (normalization.py:39-46, assembled as transformer_nova.py:87-89), MotionEmbed
(embeddings.py:119-137), VideoPosEmbed.get_time_embed (:103-111) and the 3-pass GuidanceScaler (guidance_scaler.py:37-57,
78-85). RESTATED (their classes import diffusers): the flow-matching sampler object and the model assembly, exactly as
in make_golden.py. Weights on the bf16 grid, stored as bf16 bits.
"""
import os
import sys

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from diffnext.models.diffusion_mlp import DiffusionMLP  # noqa: E402
from diffnext.models.embeddings import MaskEmbed, MotionEmbed, PosEmbed, RotaryEmbed3D, TextEmbed, VideoPosEmbed  # noqa: E402
from diffnext.models.normalization import AdaLayerNorm  # noqa: E402
from diffnext.models.vision_transformer import VisionTransformer  # noqa: E402
from make_golden import HERE, FlowMatchSampler, Model, bf16_bits, to_bf16_grid  # noqa: E402


def build_video_model(D, heads, depths, latent_hw, image_dim, patch, token_dim, token_len, rotary, base_t, mixer_rank):
    """RESTATED assembly of transformer_nova.py:73-101 (video_base_size[0] = base_t > 1 -> MotionEmbed; video_mixer_rank)."""
    hd = D // heads
    image_base = (latent_hw[0] // patch, latent_hw[1] // patch)
    video_base = (base_t, image_base[0] // 2, image_base[1] // 2)
    video_encoder = VisionTransformer(depths[0], D, heads, patch_size=patch * 2, image_size=tuple(latent_hw), image_dim=image_dim)
    image_encoder = VisionTransformer(depths[1], D, heads, patch_size=patch, image_size=tuple(latent_hw), image_dim=image_dim)
    image_decoder = DiffusionMLP(depths[2], D, cond_dim=D, patch_size=patch, image_dim=image_dim)
    if rotary:
        video_pos_embed, image_pos_embed = RotaryEmbed3D(hd, video_base[1:]), RotaryEmbed3D(hd, image_base)
    else:
        video_pos_embed, image_pos_embed = VideoPosEmbed(D, video_base), None
        image_encoder.pos_embed = PosEmbed(D, image_base)
    if mixer_rank is not None:
        video_encoder.mixer = AdaLayerNorm(D, max(mixer_rank, 0), eps=None)
    return Model(video_encoder=video_encoder, image_encoder=image_encoder, image_decoder=image_decoder,
                 mask_embed=MaskEmbed(D), text_embed=TextEmbed(token_dim, D, token_len), video_pos_embed=video_pos_embed,
                 image_pos_embed=image_pos_embed, motion_embed=MotionEmbed(D), sample_scheduler=FlowMatchSampler()).eval()


def make_case(name, seed, D, heads, depths, latent_hw, token_dim, token_len, rotary, B, K, S, T, mixer_rank, guidance=4.0,
              image_dim=3, patch=1, flow=5):
    torch.manual_seed(seed)
    m = build_video_model(D, heads, depths, latent_hw, image_dim, patch, token_dim, token_len, rotary, T, mixer_rank)
    to_bf16_grid(m, seed + 1)
    g = torch.Generator().manual_seed(1234 + seed)
    lens = [int(v) for v in torch.randint(2, token_len + 1, (B,), generator=g)]
    prompt_embeds = [(torch.randn(n, token_dim, generator=g) * 0.5).bfloat16().float() for n in lens]
    pe = m.text_embed.encode_prompts(prompt_embeds)  # RESTATED pipeline_nova.py:204-215 (prompt_embeds, guidance > 1)
    prompt = torch.cat([pe, m.text_embed.weight[: pe.shape[1]].expand(pe.shape[0], -1, -1)])
    N = (latent_hw[0] // patch) * (latent_hw[1] // patch)
    mask_len = np.round(np.cos(0.5 * np.pi * np.arange(K + 1) / K) * N).astype("int64")  # RESTATED pipeline_nova.py:129-132
    num_preds = mask_len[:-1] - mask_len[1:]
    sample_seed = 11 + seed

    def run(**extra):
        inputs = {"prompt": prompt.clone(), "num_preds": num_preds, "guidance_scale": guidance, "batch_size": B,
                  "generator": torch.Generator().manual_seed(sample_seed), "num_diffusion_steps": S, "max_latent_length": T,
                  "tqdm1": False, "tqdm2": False, "guidance_trunc": 0, "guidance_renorm": 1, "image_guidance_scale": 0,
                  "spatiotemporal_guidance_scale": 0, "motion_flow": [flow] * B}  # pipeline_nova.py:126-138
        inputs.update(extra)
        with torch.no_grad():
            return m(inputs)["x"]

    trace = {}
    hook = m.video_encoder.register_forward_hook(lambda mod, a, out: trace.setdefault("c", []).append(out.detach().clone()))
    out = run()
    hook.remove()
    order = m.mask_embed.pred_ids.clone()
    # replay of the generator draws: one uniform, then one normal per AR step of every generated frame
    gen2 = torch.Generator().manual_seed(sample_seed)
    u_dist = torch.empty(B, N, 1).uniform_(generator=gen2)
    n_steps = len([v for v in num_preds if v > 0])
    noises = [torch.empty(B, image_dim, *latent_hw).normal_(generator=gen2) for _ in range(T * n_steps)]
    assert torch.equal(u_dist.argsort(dim=1), order)

    arrays = {"w/" + k: bf16_bits(v) for k, v in m.state_dict().items()}
    meta = dict(D=D, heads=heads, video_depth=depths[0], image_depth=depths[1], decoder_depth=depths[2], latent_h=latent_hw[0],
                latent_w=latent_hw[1], image_dim=image_dim, patch=patch, token_dim=token_dim, token_len=token_len,
                rotary=int(rotary), B=B, K=K, S=S, T=T, sample_seed=sample_seed, flow=flow,
                mixer_rank=-999 if mixer_rank is None else mixer_rank)
    arrays.update({"meta/" + k: np.asarray(v) for k, v in meta.items()})
    arrays["meta/guidance"] = np.asarray(guidance, dtype="float64")
    arrays["in/prompt"] = prompt.numpy()
    for i, p_ in enumerate(prompt_embeds):
        arrays[f"in/prompt_embeds/{i}"] = p_.numpy()
    arrays["in/num_preds"] = num_preds
    arrays["in/u_dist"] = u_dist.numpy()
    arrays["in/noises"] = torch.stack(noises).numpy()
    arrays["out/x"] = out.numpy()
    arrays["out/order"] = order.numpy()
    arrays["out/c_frames"] = torch.stack(trace["c"]).numpy()  # conditioning-encoder output of every frame (before the mixer)
    # the 3-pass guidance branches on the same model and seed (guidance_scaler.py:46-57,78-85)
    arrays["out/x_image_guidance"] = run(image_guidance_scale=1.5).numpy()
    arrays["out/x_spatiotemporal_guidance"] = run(spatiotemporal_guidance_scale=0.75).numpy()
    arrays["out/x_image_guidance_renorm"] = run(image_guidance_scale=1.5, guidance_renorm=0.4, guidance_trunc=300.0).numpy()
    # image-to-video: the first frame is given, frames 1.. are generated (transformer_3d.py:159-160)
    first = out[:, :, 0].clone()
    arrays["out/x_prefilled"] = run(latents=[first]).numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez(path, **arrays)
    print(f"{name}: x {tuple(out.shape)} |x|max {out.abs().max():.4f} num_preds {num_preds.tolist()} -> {path} "
          f"({os.path.getsize(path) / 1e6:.2f} MB); 3-pass deltas: image {float((arrays['out/x_image_guidance'] - out.numpy()).__abs__().max()):.3f} "
          f"spatiotemporal {float(np.abs(arrays['out/x_spatiotemporal_guidance'] - out.numpy()).max()):.3e} "
          f"prefilled frame-1 delta {float(np.abs(arrays['out/x_prefilled'][:, :, 1] - out.numpy()[:, :, 1]).max()):.3e}")


if __name__ == "__main__":
    make_case("tiny_video_rope", 3, D=128, heads=2, depths=(2, 2, 1), latent_hw=(8, 8), token_dim=64, token_len=8, rotary=True,
              B=2, K=3, S=2, T=3, mixer_rank=-1)
    make_case("tiny_video_abspe", 4, D=128, heads=2, depths=(1, 2, 1), latent_hw=(4, 8), token_dim=64, token_len=8, rotary=False,
              B=1, K=3, S=2, T=2, mixer_rank=None)
