"""CPU: the drop-in `diffnext` package (PyTorch definitions + host logic of the pipeline) against
the golden vectors of the reference, and its API surface (names, kwargs, state_dict schema)."""
import inspect
import os
import sys

import numpy as np
import pytest
import torch

from golden_util import CASES, VIDEO_CASES, Golden

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nova_pointcloud_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)  # how a user shadows the reference's `diffnext`

from diffnext.models.transformers import transformer_nova as TN  # noqa: E402
from diffnext.pipelines import NOVAPipeline  # noqa: E402
from diffnext.pipelines.nova.pipeline_nova import cosine_set_sizes, points_from_latents  # noqa: E402
from diffnext.schedulers import DDPMScheduler, FlowMatchEulerDiscreteScheduler  # noqa: E402


def build_from_golden(g, dtype=torch.float32, device="cpu"):
    """NOVATransformer3DModel with the fixture's tiny architecture, weights loaded strictly."""
    m = g.meta
    D, heads = m["D"], m["heads"]
    TN.VIDEO_ENCODERS.register(f"vit_d{m['video_depth']}w{D}", TN._vit, depth=m["video_depth"], embed_dim=D, num_heads=heads)
    TN.IMAGE_ENCODERS.register(f"vit_d{m['image_depth']}w{D}", TN._vit, depth=m["image_depth"], embed_dim=D, num_heads=heads)
    TN.IMAGE_DECODERS.register(f"mlp_d{m['decoder_depth']}w{D}", TN._mlp, depth=m["decoder_depth"], embed_dim=D)
    stride = 16 // m["patch"]  # patch = 15 // stride + 1
    H, W = m["latent_h"], m["latent_w"]
    base = [H // m["patch"], W // m["patch"]]
    model = TN.NOVATransformer3DModel(
        image_dim=m["image_dim"], image_size=(H * stride, W * stride), image_stride=stride, text_token_dim=m["token_dim"],
        text_token_len=m["token_len"], image_base_size=base, video_base_size=[m.get("T", 1), base[0] // 2, base[1] // 2],
        video_mixer_rank=None if m.get("mixer_rank", -999) == -999 else m["mixer_rank"], rotary_pos_embed=bool(m["rotary"]),
        arch=(f"vit_d{m['video_depth']}w{D}", f"vit_d{m['image_depth']}w{D}", f"mlp_d{m['decoder_depth']}w{D}"))
    missing = model.load_state_dict(g.weights, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model.to(device=device, dtype=dtype).eval()


@pytest.fixture(scope="module", params=CASES)
def gold(request):
    return Golden(request.param)


def test_state_dict_schema_matches_reference(gold):
    """Same parameter / persistent-buffer names and shapes as the reference's modules produced."""
    model = build_from_golden(gold)
    sd = model.state_dict()
    assert set(sd) == set(gold.weights)
    assert all(tuple(sd[k].shape) == tuple(gold.weights[k].shape) for k in sd)


def test_pipeline_cpu_matches_reference(gold):
    """NOVAPipeline.__call__ on CPU from the same seed reproduces the reference's latent bit-for-bit-ish."""
    m = gold.meta
    pipe = NOVAPipeline(transformer=build_from_golden(gold), scheduler=FlowMatchEulerDiscreteScheduler())
    out = pipe(prompt_embeds=gold.prompt_embeds, num_inference_steps=m["K"], num_diffusion_steps=m["S"],
               guidance_scale=m["guidance"], generator=torch.Generator().manual_seed(m["sample_seed"]),
               output_type="latent", disable_progress_bar=True)
    x, ref = out.frames, gold.t["out/x"]
    assert "frames" in out and out["frames"] is x and x.shape == ref.shape
    assert (x - ref).abs().max() <= 1e-5 * ref.abs().max()
    assert torch.equal(pipe.transformer.mask_embed.pred_ids, gold.t["out/order"])
    pts = points_from_latents(x)
    assert pts.shape == (m["B"], m["latent_h"] * m["latent_w"], 3)


@pytest.fixture(scope="module", params=VIDEO_CASES)
def vgold(request):
    return Golden(request.param)


def video_call(vgold, device="cpu", dtype=torch.float32, **kw):
    m = vgold.meta
    pipe = NOVAPipeline(transformer=build_from_golden(vgold, dtype, device), scheduler=FlowMatchEulerDiscreteScheduler())
    args = dict(prompt_embeds=[p.to(device) for p in vgold.prompt_embeds], num_inference_steps=m["K"], num_diffusion_steps=m["S"],
                max_latent_length=m["T"], guidance_scale=m["guidance"], motion_flow=m["flow"],
                generator=torch.Generator().manual_seed(m["sample_seed"]), output_type="latent", disable_progress_bar=True)
    args.update(kw)
    return pipe, pipe(**args).frames


VIDEO_VARIANTS = [("out/x", {}), ("out/x_image_guidance", dict(image_guidance_scale=1.5)),
                  ("out/x_spatiotemporal_guidance", dict(spatiotemporal_guidance_scale=0.75)),
                  ("out/x_image_guidance_renorm", dict(image_guidance_scale=1.5, guidance_renorm=0.4, guidance_trunc=300.0))]


@pytest.mark.parametrize("key,kw", VIDEO_VARIANTS)
def test_multi_frame_pipeline_cpu_matches_reference(vgold, key, kw):
    """max_latent_length > 1 (KV-cached conditioning encoder, frame mixer, motion tokens) and the 3-pass guidance forms,
    module path on CPU against the reference's generate_video runs."""
    pipe, x = video_call(vgold, **kw)
    ref = vgold.t[key]
    assert x.shape == ref.shape
    assert (x - ref).abs().max() <= 2e-5 * ref.abs().max()
    sd = pipe.transformer.state_dict()
    assert set(sd) == set(vgold.weights)


def test_prefilled_first_frame_cpu_matches_reference(vgold):
    _, x = video_call(vgold, latents=[vgold.t["out/x"][:, :, 0].clone()])
    ref = vgold.t["out/x_prefilled"]
    assert (x - ref).abs().max() <= 2e-5 * ref.abs().max()


def test_call_signature_matches_reference():
    """Keyword set and defaults of NOVAPipeline.__call__ (reference pipeline_nova.py:55-78)."""
    sig = inspect.signature(NOVAPipeline.__call__)
    want = dict(prompt=None, num_inference_steps=64, num_diffusion_steps=25, max_latent_length=1, guidance_scale=5,
                guidance_trunc=0, guidance_renorm=1, image_guidance_scale=0, spatiotemporal_guidance_scale=0,
                flow_shift=None, motion_flow=5, negative_prompt=None, image=None, num_images_per_prompt=1,
                generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None,
                disable_progress_bar=False, output_type="pil")
    got = {k: v.default for k, v in sig.parameters.items() if k not in ("self", "kwargs")}
    assert got == want
    init = inspect.signature(NOVAPipeline.__init__)
    assert list(init.parameters)[1:] == ["transformer", "scheduler", "vae", "text_encoder", "tokenizer", "trust_remote_code"]


def test_registry_names_and_errors():
    for w, h in ((768, 12), (1024, 16), (1536, 16)):
        assert TN.VIDEO_ENCODERS.has(f"vit_d16w{w}") and TN.IMAGE_ENCODERS.has(f"vit_d32w{w}") and TN.IMAGE_DECODERS.has(f"mlp_d6w{w}")
    assert TN.IMAGE_DECODERS.has("mlp_d3w1280")
    with pytest.raises(KeyError):
        TN.IMAGE_ENCODERS.get("vit_d32w999")


def test_schedule_and_sampler_grid():
    assert cosine_set_sizes(256, 4).tolist() == [19, 56, 83, 98]
    s = FlowMatchEulerDiscreteScheduler()
    s.set_timesteps(25)
    assert len(s.timesteps) == 25 and len(s.sigmas) == 26 and s.sigmas[-1] == 0
    x, v = torch.ones(2, 3), torch.full((2, 3), 2.0)
    s._step_index = None
    out = s.step(v, s.timesteps[0], x).prev_sample
    assert torch.allclose(out, x + (s.sigmas[1] - s.sigmas[0]) * v)
    s2 = FlowMatchEulerDiscreteScheduler(shift=3.0)
    s2.set_timesteps(8)
    assert all(a > b for a, b in zip(s2.sigmas[:-1], s2.sigmas[1:]))


def test_ddpm_scheduler_steps():
    s = DDPMScheduler(num_train_timesteps=100, clip_sample=False)
    s.set_timesteps(10)
    assert s.timesteps.tolist() == list(range(90, -1, -10))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 4, 4, generator=g)
    for t in s.timesteps:
        x = s.step(torch.zeros_like(x), t, x, generator=g).prev_sample
    assert torch.isfinite(x).all()


def test_ddpm_scheduler_and_engine_plan_hit_the_hand_derived_known_answer():
    """The literals of tests/test_oracle.py::test_ddpm_known_answer_derived_by_hand_from_the_source_text (2 steps over 4 training
    timesteps, linear betas 0.1 .. 0.4, worked out by hand from scheduling_ddpm.py:143-146,196-199,268-305,319-325) against the
    drop-in's DDPMScheduler class and the per-step plan the HIP engine hands to nova_sampler_step."""
    from test_oracle import DDPM_KNOWN_ANSWER as want

    kw = dict(num_train_timesteps=4, beta_start=0.1, beta_end=0.4)
    # the drop-in's scheduler class: one step from known x, eps with the noise replaced by a constant field
    from nova_pointcloud_amd.engine import sampler_plan

    sch = DDPMScheduler(clip_sample=False, **kw)
    sch.set_timesteps(2)
    assert [int(t) for t in sch.timesteps] == [2, 0]
    x, eps = torch.full((1, 1, 2, 2), 0.5), torch.full((1, 1, 2, 2), -0.25)
    kx, kv, c0, cx, sg, _ = want[2]
    mean2 = c0 * (kx * 0.5 + kv * -0.25) + cx * 0.5  # = 0.876901... by the literals above
    draws = torch.Generator().manual_seed(3)
    noise = torch.randn(1, 1, 2, 2, generator=torch.Generator().manual_seed(3))
    got = sch.step(eps, sch.timesteps[0], x.clone(), generator=draws).prev_sample
    assert (got - (mean2 + sg * noise)).abs().max() <= 2e-6
    last = sch.step(eps, sch.timesteps[1], x.clone(), generator=draws).prev_sample  # t = 0: the predicted x0, no noise
    assert (last - (want[0][0] * 0.5 + want[0][1] * -0.25)).abs().max() <= 2e-6
    ts, coefs, ancestral = sampler_plan(sch, 2)  # what nova_sampler_step receives per step: (kx, kv, clip, c0, cx, sigma)
    assert ancestral and [int(v) for v in ts] == [2, 0]
    for (kx_, kv_, clip, c0_, cx_, sg_), t in zip(coefs, (2, 0)):
        w = want[t]
        assert clip == 0.0 and max(abs(kx_ - w[0]), abs(kv_ - w[1]), abs(c0_ - w[2]), abs(cx_ - w[3]), abs(sg_ - w[4])) <= 2e-6


def test_save_and_load_roundtrip(gold, tmp_path):
    model = build_from_golden(gold)
    pipe = NOVAPipeline(transformer=model, scheduler=FlowMatchEulerDiscreteScheduler(shift=2.0))
    pipe.save_pretrained(str(tmp_path))
    again = NOVAPipeline.from_pretrained(str(tmp_path))
    assert again.scheduler.config.shift == 2.0
    for (k, a), (_, b) in zip(model.state_dict().items(), again.transformer.state_dict().items()):
        assert torch.equal(a, b), k


def test_pipeline_directory_layout_and_builder(gold, tmp_path):
    """On-disk layout (reference pipelines/builder.py:31-125): model_index.json + component
    directories, component replacement / removal / config override through get_pipeline_path, diffusers-style index
    entries of libraries that are not installed."""
    import json

    from diffnext.pipelines.builder import build_diffusion_scheduler, build_pipeline, get_pipeline_path

    base, other = tmp_path / "base", tmp_path / "other"
    model = build_from_golden(gold)
    NOVAPipeline(transformer=model, scheduler=FlowMatchEulerDiscreteScheduler(shift=2.0)).save_pretrained(str(base))
    index = json.load(open(base / "model_index.json"))
    assert index["_class_name"] == "NOVAPipeline" and index["transformer"][1] == "NOVATransformer3DModel"
    assert (base / "transformer" / "config.json").exists() and (base / "scheduler" / "scheduler_config.json").exists()
    # a checkpoint as published: VAE / text encoder entries of libraries this box does not have
    index.update(vae=["diffusers", "AutoencoderKL"], text_encoder=["transformers", "NoSuchTextModel"])
    json.dump(index, open(base / "model_index.json", "w"))
    pipe = build_pipeline(str(base), NOVAPipeline, dtype=torch.float32)
    assert pipe.vae is None and pipe.text_encoder is None and pipe.scheduler.config.shift == 2.0
    # replace the transformer, drop the VAE, override the transformer's config (module_dict / module_config of configs/*.yaml)
    other_model = build_from_golden(gold)
    with torch.no_grad():
        other_model.mask_embed.mask_token.add_(1.0)
    other_model.save_pretrained(str(other))
    assert get_pipeline_path(str(base)) == str(base)
    cfg = json.load(open(other / "config.json"))
    path = get_pipeline_path(str(base), module_dict={"transformer": str(other), "vae": ""}, module_config={"scheduler": None})
    idx2 = json.load(open(os.path.join(path, "model_index.json")))
    assert "vae" not in idx2 and "transformer" in idx2
    assert os.path.islink(os.path.join(path, "scheduler", "scheduler_config.json"))
    loaded = NOVAPipeline.from_pretrained(path)
    assert torch.equal(loaded.transformer.mask_embed.mask_token, model.mask_embed.mask_token)  # base dir was linked first
    path2 = get_pipeline_path(str(other.parent / "other_as_pipe") if False else str(base), module_config={"transformer": cfg},
                              target_path=str(tmp_path / "t2"))
    assert json.load(open(os.path.join(path2, "transformer", "config.json")))["arch"] == cfg["arch"]
    assert not os.path.islink(os.path.join(path2, "transformer", "config.json"))
    # schedulers by directory config and by object
    sched_dir = tmp_path / "sched"
    os.makedirs(sched_dir)
    json.dump({"_noise_class_name": "FlowMatchEulerDiscreteScheduler", "_sample_class_name": "DDPMScheduler", "shift": 1.0,
               "num_train_timesteps": 1000}, open(sched_dir / "scheduler_config.json", "w"))
    assert type(build_diffusion_scheduler(str(sched_dir))).__name__ == "FlowMatchEulerDiscreteScheduler"
    assert type(build_diffusion_scheduler(str(sched_dir), sample=True)).__name__ == "DDPMScheduler"
    clone = build_diffusion_scheduler(FlowMatchEulerDiscreteScheduler(shift=3.0))
    assert clone.config.shift == 3.0 and build_diffusion_scheduler(None) is None


def test_guidance_le_one_runs_single_pass(gold):
    m = gold.meta
    pipe = NOVAPipeline(transformer=build_from_golden(gold), scheduler=FlowMatchEulerDiscreteScheduler())
    out = pipe(prompt_embeds=gold.prompt_embeds, num_inference_steps=2, num_diffusion_steps=2, guidance_scale=1,
               generator=torch.Generator().manual_seed(0), output_type="latent", disable_progress_bar=True)
    assert out.frames.shape[0] == m["B"] and torch.isfinite(out.frames).all()


def test_pipeline_cpu_trunc_and_renorm_match_reference(gold):
    m = gold.meta
    pipe = NOVAPipeline(transformer=build_from_golden(gold), scheduler=FlowMatchEulerDiscreteScheduler())
    out = pipe(prompt_embeds=gold.prompt_embeds, num_inference_steps=m["K"], num_diffusion_steps=m["S"],
               guidance_scale=m["guidance"], guidance_trunc=450.0, guidance_renorm=0.3,
               generator=torch.Generator().manual_seed(m["sample_seed"]), output_type="latent", disable_progress_bar=True)
    ref = gold.t["out/x_trunc450_renorm03"]
    assert (out.frames - ref).abs().max() <= 1e-5 * ref.abs().max()


def test_pipeline_cpu_ddpm_matches_oracle(gold):
    """Two independent restatements of scheduling_ddpm.py (the scheduler class here, oracle.ddpm_plan) agree."""
    from oracle import nova_oracle as O

    m = gold.meta
    kw = dict(num_train_timesteps=1000, beta_schedule="scaled_linear", beta_start=0.00085, beta_end=0.012, prediction_type="epsilon")
    pipe = NOVAPipeline(transformer=build_from_golden(gold), scheduler=DDPMScheduler(**kw))
    out = pipe(prompt_embeds=gold.prompt_embeds, num_inference_steps=m["K"], num_diffusion_steps=4, guidance_scale=m["guidance"],
               generator=torch.Generator().manual_seed(21), output_type="latent", disable_progress_bar=True).frames
    ref = O.generate(gold.weights, gold.oracle_config(), gold.t["in/prompt"], gold.t["in/num_preds"].numpy(), num_diffusion_steps=4,
                     guidance_scale=m["guidance"], generator=torch.Generator().manual_seed(21), ddpm=kw)
    assert (out - ref).abs().max() <= 2e-5 * ref.abs().max()


def test_training_step_matches_reference(gold):
    """SURVEY section 8f N2 (training through the torch definition of the same modules): one forward/backward of
    `train_video` (reference transformer_3d.py:166-190, get_losses :79-100) under the recorded torch / numpy global
    seeds reproduces the reference's loss, mask and parameter gradients. The fixture was produced by the reference's
    own Transformer3DModel with the training side of its flow-matching scheduler restated (make_golden.py)."""
    import numpy as np

    model = build_from_golden(gold)
    model.noise_scheduler = FlowMatchEulerDiscreteScheduler()
    model.train()
    seed = int(gold.t["train/seed"])
    torch.manual_seed(seed)
    np.random.seed(seed)
    out = model({"x": gold.t["train/x"].clone(), "prompt": [p.clone() for p in gold.prompt_embeds]})
    loss = out["loss"]
    ref = float(gold.t["train/loss"])
    assert abs(float(loss.detach()) - ref) <= 1e-5 * abs(ref)
    assert torch.equal(model.mask_embed.mask.detach().float(), gold.t["train/mask"].float())
    loss.backward()
    params = dict(model.named_parameters())
    checked = 0
    for key, want in gold.t.items():
        if not key.startswith("train/grad/"):
            continue
        got = params[key[len("train/grad/"):]].grad
        assert got is not None and got.shape == want.shape
        assert (got - want).abs().max() <= 2e-5 * want.abs().max() + 1e-9
        checked += 1
    assert checked >= 5


def test_training_attention_dispatch_rules():
    """Host logic of the training hook (nova_pointcloud_amd/autograd.py): only bf16 device tensors [S, heads, L, 64 | 96]
    without a mask go to the HIP attention; everything else keeps F.scaled_dot_product_attention. On this CPU-only side
    nothing may qualify, and `Attention.forward` with grad enabled must run (and differentiate) without the library."""
    from diffnext.models.vision_transformer import Attention
    from nova_pointcloud_amd import autograd as A

    q = torch.zeros(2, 3, 16, 64, dtype=torch.bfloat16)
    assert not A.attention_supported(q)                                   # CPU tensor
    meta = torch.zeros(2, 3, 16, 64, dtype=torch.bfloat16, device="meta")
    assert not A.attention_supported(meta)                                # not a CUDA device
    before = A.stats["attention_calls"]
    attn = Attention(128, 2)
    x = torch.randn(2, 10, 128, requires_grad=True)
    attn(x).square().mean().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and A.stats["attention_calls"] == before

    class FakeCuda(object):  # shape / dtype / mask gating without a device
        is_cuda = True

        def __init__(self, shape, dtype):
            self.shape, self.dtype = shape, dtype

        def dim(self):
            return len(self.shape)

    ok = FakeCuda((2, 3, 16, 64), torch.bfloat16)
    assert A.attention_supported(ok) == A._ENABLED and A.attention_supported(FakeCuda((2, 3, 16, 96), torch.bfloat16)) == A._ENABLED
    assert not A.attention_supported(ok, attn_mask=torch.zeros(16, 16))
    assert not A.attention_supported(FakeCuda((2, 3, 16, 64), torch.float32))
    assert not A.attention_supported(FakeCuda((2, 3, 16, 128), torch.bfloat16))
    assert not A.attention_supported(FakeCuda((6, 16, 64), torch.bfloat16))


def test_config0_in_full_on_the_cpu_module_path_matches_oracle():
    """BASELINE configs[0] IN FULL - NOVA-d48w768 (16 + 32 ViT blocks, 6 diffusion-MLP blocks, random init), 256 points,
    4 AR x 4 diffusion steps, batch 1 - through `NOVAPipeline.__call__` on the drop-in's CPU module path (the reference's
    own CPU/PyTorch semantics of the module API), against the oracle on the same weights, prompt and host generator seed.
    The same oracle run is what the HIP path is held to on the GPU box (tests/test_gpu_parity_full.py, case
    config0_d48w768_256pts_K4S4)."""
    import bench
    from oracle import nova_oracle as O

    H = W = 16
    K = S = 4
    threads = torch.get_num_threads()
    torch.set_num_threads(max(threads, min(bench.host_cores(), 8)))
    try:
        pipe = bench.build_pipeline(768, 12, H, W, torch.float32, torch.device("cpu"))
        sd = {k: v.detach().clone() for k, v in pipe.transformer.state_dict().items()}
        prompts = bench.synthetic_prompts(1, "cpu", torch.float32, seed=4321)
        sched = [int(v) for v in O.cosine_schedule(H * W, K) if v > 0]
        assert sched == [19, 56, 83, 98]
        cfg = O.make_config(3, (H, W), 1, 768, 12, 16, 32, 6, 256, rotary=True)
        prompt = O.encode_prompt_embeds(sd["text_embed.weight"], prompts, 256)
        with torch.no_grad():
            ref = O.generate(sd, cfg, prompt, sched, num_diffusion_steps=S, guidance_scale=5.0, generator=torch.Generator().manual_seed(29))
            out = pipe(prompt_embeds=prompts, num_inference_steps=K, num_diffusion_steps=S, guidance_scale=5, output_type="latent",
                       disable_progress_bar=True, generator=torch.Generator().manual_seed(29)).frames
    finally:
        torch.set_num_threads(threads)
    assert tuple(out.shape) == (1, 3, 1, H, W) and torch.isfinite(out).all()
    err = ((out.double() - ref.double()).abs().max() / ref.double().abs().max()).item()
    assert err < 1e-4, err  # two f32 CPU evaluations of the same arithmetic: summation order only
