"""Generate the golden vectors in tests/golden/*.npz by RUNNING THE REFERENCE'S OWN MODULES.

Run in the build container only (needs /root/reference, read-only):

    python tests/golden/make_golden.py

What runs from the reference, imported as-is (no shims): VisionTransformer, DiffusionMLP,
MaskEmbed, TextEmbed, RotaryEmbed3D, PosEmbed, VideoPosEmbed and the whole AR generator
Transformer3DModel.forward/generate_video/generate_frame/denoise
(diffnext/models/{vision_transformer,diffusion_mlp,embeddings}.py,
diffnext/models/transformers/transformer_3d.py). These hold all arithmetic of the hot path.

What cannot be imported here (`diffusers` is absent; it stays absent): the scheduler classes,
NOVATransformer3DModel and NOVAPipeline. The pieces of them the loop needs are re-stated below
from their source text and are marked RESTATED: the flow-matching sampler object
(scheduling_cfm.py:92-104,125-140), the model assembly (transformer_nova.py:73-101), the cosine
schedule and the prompt_embeds/negative handling (pipeline_nova.py:129-132,204-215).

Weights are default-initialised by the reference constructors under a fixed seed, biases / norm
affines are perturbed (defaults are 0 / 1 and would hide bugs), and everything is rounded to
the bf16 grid so one fixture serves the f32 and the bf16 tests (stored as uint16 bf16 bits).
"""
import os
import sys

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")  # embeddings.py:36 torch.compile -> eager (same math, no 35 s compile)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from diffnext.models.diffusion_mlp import DiffusionMLP  # noqa: E402
from diffnext.models.embeddings import MaskEmbed, PosEmbed, RotaryEmbed3D, TextEmbed, VideoPosEmbed  # noqa: E402
from diffnext.models.transformers.transformer_3d import Transformer3DModel  # noqa: E402
from diffnext.models.vision_transformer import VisionTransformer  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


class FlowMatchSampler(object):
    """RESTATED from scheduling_cfm.py (class needs diffusers): only what Transformer3DModel touches."""

    def __init__(self, num_train_timesteps=1000, shift=1.0):
        self.n, self.shift = num_train_timesteps, shift
        t = np.arange(1, self.n + 1, dtype="float32")[::-1]
        s = t / self.n
        s = shift * s / (1 + (shift - 1) * s)
        self.sigma_min, self.sigma_max = float(s[-1]), float(s[0])
        self._step_index = None

    def set_timesteps(self, num_inference_steps):
        t = np.linspace(self.sigma_max * self.n, self.sigma_min * self.n, num_inference_steps, dtype="float32")
        s = t / self.n
        s = self.shift * s / (1 + (self.shift - 1) * s)
        self.sigmas, self.timesteps, self._step_index = s.tolist() + [0], s * self.n, None

    def step(self, model_output, timestep, sample, generator=None):
        if self._step_index is None:
            self._step_index = 0
        dt = self.sigmas[self._step_index + 1] - self.sigmas[self._step_index]
        self._step_index += 1
        return type("Out", (), {"prev_sample": model_output.mul(dt).add_(sample)})


class FlowMatchTrainer(object):
    """RESTATED from scheduling_cfm.py:40-49,92-95,111-123 (class needs diffusers): the training side that
    Transformer3DModel.get_losses touches (sample_timesteps, add_noise, .timestep, .config.prediction_type absent)."""

    def __init__(self, num_train_timesteps=1000, shift=1.0):
        t = np.arange(1, num_train_timesteps + 1, dtype="float32")[::-1]
        s = t / num_train_timesteps
        s = shift * s / (1 + (shift - 1) * s)
        self.n = num_train_timesteps
        self.timesteps, self.sigmas = torch.as_tensor(s * num_train_timesteps), torch.as_tensor(s.copy())
        self.timestep = self.sigma = None
        self.config = type("Cfg", (), {})()

    def sample_timesteps(self, size, device=None):
        dist = torch.normal(0, 1, size, device=device).sigmoid_()
        return dist.mul_(self.n).to(dtype=torch.int64)

    def add_noise(self, original_samples, noise, timesteps):
        dtype, device = original_samples.dtype, original_samples.device
        self.timestep = self.timesteps.to(device=device)[timesteps]
        self.sigma = self.sigmas.to(device=device, dtype=dtype)[timesteps]
        self.sigma = self.sigma.view(timesteps.shape + (1,) * (noise.dim() - timesteps.dim()))
        return self.sigma * noise + (1.0 - self.sigma) * original_samples


TRAIN_GRADS = ["mask_embed.mask_token", "image_decoder.head.weight", "image_encoder.blocks.0.attn.qkv.weight",
               "video_encoder.blocks.0.mlp.fc1.bias", "text_embed.proj.weight", "image_decoder.blocks.0.norm1.proj.weight"]


def train_record(m, seed, B, image_dim, latent_hw, prompt_embeds):
    """One training forward/backward of the reference (Transformer3DModel.train_video, transformer_3d.py:166-190):
    loss and a few parameter gradients under fixed torch / numpy global seeds (MaskEmbed draws its ratio from
    scipy.stats.truncnorm, TextEmbed's prompt dropout from np.random; everything else from torch's global RNG)."""
    m.noise_scheduler = FlowMatchTrainer()
    m.train()
    g = torch.Generator().manual_seed(4321 + seed)
    x = (torch.randn(B, image_dim, *latent_hw, generator=g) * 0.8).bfloat16().float()
    torch.manual_seed(100 + seed)
    np.random.seed(100 + seed)
    m.zero_grad(set_to_none=True)
    out = m({"x": x.clone(), "prompt": [p.clone() for p in prompt_embeds]})
    out["loss"].backward()
    grads = dict(m.named_parameters())
    rec = {"train/x": x.numpy(), "train/loss": out["loss"].detach().double().numpy(),
           "train/seed": np.asarray(100 + seed), "train/mask": m.mask_embed.mask.detach().numpy()}
    for k in TRAIN_GRADS:
        if k in grads and grads[k].grad is not None:
            rec["train/grad/" + k] = grads[k].grad.detach().numpy().copy()
    m.zero_grad(set_to_none=True)
    m.eval()
    return rec


class Model(Transformer3DModel):
    """Transformer3DModel + the two properties diffusers' ModelMixin would provide."""

    dtype, device = torch.float32, torch.device("cpu")


def build_model(D, heads, depths, latent_hw, image_dim, patch, token_dim, token_len, rotary):
    """RESTATED assembly of transformer_nova.py:73-101 on the reference's own layer classes."""
    hd = D // heads
    image_size = tuple(latent_hw)
    image_base = (latent_hw[0] // patch, latent_hw[1] // patch)
    video_base = (1, image_base[0] // 2, image_base[1] // 2)
    video_encoder = VisionTransformer(depths[0], D, heads, patch_size=patch * 2, image_size=image_size, image_dim=image_dim)
    image_encoder = VisionTransformer(depths[1], D, heads, patch_size=patch, image_size=image_size, image_dim=image_dim)
    image_decoder = DiffusionMLP(depths[2], D, cond_dim=D, patch_size=patch, image_dim=image_dim)
    if rotary:
        video_pos_embed, image_pos_embed = RotaryEmbed3D(hd, video_base[1:]), RotaryEmbed3D(hd, image_base)
    else:
        video_pos_embed, image_pos_embed = VideoPosEmbed(D, video_base), None
        image_encoder.pos_embed = PosEmbed(D, image_base)
    return Model(video_encoder=video_encoder, image_encoder=image_encoder, image_decoder=image_decoder,
                 mask_embed=MaskEmbed(D), text_embed=TextEmbed(token_dim, D, token_len),
                 video_pos_embed=video_pos_embed, image_pos_embed=image_pos_embed,
                 sample_scheduler=FlowMatchSampler()).eval()


def to_bf16_grid(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if name.endswith("bias"):
                prm.add_(torch.randn(prm.shape, generator=g) * 0.05)
            elif ".norm" in name and name.endswith("weight") and prm.dim() == 1:
                prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
        for t in model.state_dict().values():  # parameters + persistent buffers only (NOT the RoPE scale buffers)
            if t.is_floating_point():
                t.copy_(t.bfloat16().float())


def bf16_bits(t):
    return t.detach().bfloat16().view(torch.int16).numpy().view(np.uint16)


def make_case(name, seed, D, heads, depths, latent_hw, token_dim, token_len, rotary, B, K, S, guidance=5.0,
              image_dim=3, patch=1):
    torch.manual_seed(seed)
    m = build_model(D, heads, depths, latent_hw, image_dim, patch, token_dim, token_len, rotary)
    to_bf16_grid(m, seed + 1)
    g = torch.Generator().manual_seed(1234 + seed)
    lens = [int(v) for v in torch.randint(2, token_len + 1, (B,), generator=g)]
    prompt_embeds = [(torch.randn(n, token_dim, generator=g) * 0.5).bfloat16().float() for n in lens]

    # the training record first: tensors the reference caches during generation are inference tensors and cannot
    # enter autograd afterwards (abs-PE table); generation below does not depend on the global RNG state
    train_rec = train_record(m, seed, B, image_dim, latent_hw, prompt_embeds)

    # RESTATED pipeline_nova.py:204-215 (prompt_embeds path, guidance > 1) on the reference's encode_prompts
    pe = m.text_embed.encode_prompts(prompt_embeds)
    neg = m.text_embed.weight[: pe.shape[1]].expand(pe.shape[0], -1, -1)
    prompt = torch.cat([pe, neg])
    # RESTATED pipeline_nova.py:129-132
    N = (latent_hw[0] // patch) * (latent_hw[1] // patch)
    ratios = np.cos(0.5 * np.pi * np.arange(K + 1) / K)
    mask_len = np.round(ratios * N).astype("int64")
    num_preds = mask_len[:-1] - mask_len[1:]

    trace = {"z": []}
    hooks = [
        m.image_encoder.register_forward_hook(lambda mod, a, out: trace["z"].append(out.detach().clone())),
        m.video_encoder.register_forward_hook(lambda mod, a, out: trace.__setitem__("c", out.detach().clone())),
    ]
    dec_calls = []
    hooks.append(m.image_decoder.register_forward_hook(
        lambda mod, a, out: dec_calls.append((a, out.detach().clone())) if len(dec_calls) < 1 else None))
    sample_seed = 7 + seed
    gen = torch.Generator().manual_seed(sample_seed)
    inputs = {"prompt": prompt.clone(), "num_preds": num_preds, "guidance_scale": guidance, "generator": gen,
              "batch_size": B, "num_diffusion_steps": S, "max_latent_length": 1, "tqdm1": False, "tqdm2": False,
              "guidance_trunc": 0, "guidance_renorm": 1, "image_guidance_scale": 0, "spatiotemporal_guidance_scale": 0}
    with torch.no_grad():
        out = m(inputs)["x"]
    [h.remove() for h in hooks]
    order = m.mask_embed.pred_ids.clone()

    # replay of the generator draws (the run above consumed exactly these, in this order)
    gen2 = torch.Generator().manual_seed(sample_seed)
    u_dist = torch.empty(B, N, 1).uniform_(generator=gen2)
    steps = [int(v) for v in num_preds if v > 0]
    noises = [torch.empty(B, image_dim, *latent_hw).normal_(generator=gen2) for _ in steps]
    assert torch.equal(u_dist.argsort(dim=1), order), "generator replay does not match the reference's draw order"

    # second run of the SAME model and seed with guidance truncation + renormalisation (guidance_scaler.py:59-72)
    gen3 = torch.Generator().manual_seed(sample_seed)
    inputs2 = dict(inputs, prompt=prompt.clone(), generator=gen3, guidance_trunc=450.0, guidance_renorm=0.3)
    inputs2.pop("c", None), inputs2.pop("x", None), inputs2.pop("latents", None)
    with torch.no_grad():
        out2 = m(inputs2)["x"]

    (dx, dt, dz, dids), dout = dec_calls[0]
    sd = {k: v for k, v in m.state_dict().items()}
    arrays = {"w/" + k: bf16_bits(v) for k, v in sd.items()}
    meta = dict(D=D, heads=heads, video_depth=depths[0], image_depth=depths[1], decoder_depth=depths[2],
                latent_h=latent_hw[0], latent_w=latent_hw[1], image_dim=image_dim, patch=patch, token_dim=token_dim,
                token_len=token_len, rotary=int(rotary), B=B, K=K, S=S, sample_seed=sample_seed)
    arrays.update({"meta/" + k: np.asarray(v) for k, v in meta.items()})
    arrays["meta/guidance"] = np.asarray(guidance, dtype="float64")
    arrays["in/prompt"] = prompt.numpy()
    for i, p_ in enumerate(prompt_embeds):
        arrays[f"in/prompt_embeds/{i}"] = p_.numpy()
    arrays["in/num_preds"] = num_preds
    arrays["in/u_dist"] = u_dist.numpy()
    arrays["in/noises"] = torch.stack(noises).numpy()
    arrays["out/x"] = out.numpy()
    arrays["out/x_trunc450_renorm03"] = out2.numpy()
    arrays["out/order"] = order.numpy()
    arrays["out/c"] = trace["c"].numpy()
    arrays["out/z_first"] = trace["z"][0].numpy()
    arrays["out/z_last"] = trace["z"][-1].numpy()
    arrays["dec/x"], arrays["dec/t"], arrays["dec/z"] = dx.numpy(), dt.numpy(), dz.numpy()
    arrays["dec/pred_ids"], arrays["dec/out"] = dids.numpy(), dout.numpy()
    arrays.update(train_rec)
    path = os.path.join(HERE, name + ".npz")
    np.savez(path, **arrays)
    print(f"{name}: train loss {float(arrays['train/loss']):.6f}; x {tuple(out.shape)} |x|max {out.abs().max():.4f} num_preds {num_preds.tolist()} -> {path} "
          f"({os.path.getsize(path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    make_case("tiny_rope", 0, D=128, heads=2, depths=(2, 2, 2), latent_hw=(8, 16), token_dim=64, token_len=8,
              rotary=True, B=2, K=5, S=3)
    make_case("tiny_abspe", 1, D=128, heads=2, depths=(2, 2, 2), latent_hw=(8, 8), token_dim=64, token_len=8,
              rotary=False, B=1, K=4, S=4)
