"""CPU: the oracle (oracle/nova_oracle.py) against the golden vectors produced by the reference's
own modules, plus hand-derived pins for the pieces restated from source text only."""
import numpy as np
import pytest
import torch

from golden_util import CASES, VIDEO_CASES, Golden
from oracle import nova_oracle as O


@pytest.fixture(scope="module", params=CASES)
def gold(request):
    return Golden(request.param)


def test_schedule_known_values():
    # values obtained with the reference pipeline; sum always = N
    assert O.cosine_schedule(256, 4).tolist() == [19, 56, 83, 98]
    s = O.cosine_schedule(2048, 64)
    assert s.sum() == 2048 and len(s) == 64 and s.min() >= 1 and s.max() == 51
    s = O.cosine_schedule(256, 64)
    assert s.sum() == 256 and (s > 0).sum() == 61


def test_cfm_sigmas_shape_and_ends():
    t, sig = O.cfm_sigmas(25)
    assert len(t) == 25 and len(sig) == 26 and sig[-1] == 0
    assert abs(sig[0] - 1.0) < 1e-7 and abs(sig[24] - 0.001) < 1e-7
    assert t.dtype == np.float32 and abs(float(t[0]) - 1000.0) < 1e-3
    t3, sig3 = O.cfm_sigmas(8, shift=3.0)
    assert all(a > b for a, b in zip(sig3[:-1], sig3[1:]))


def test_golden_schedule_matches(gold):
    m = gold.meta
    N = (m["latent_h"] // m["patch"]) * (m["latent_w"] // m["patch"])
    assert O.cosine_schedule(N, m["K"]).tolist() == gold.t["in/num_preds"].tolist()


def test_prompt_encoding_matches_reference(gold):
    got = O.encode_prompt_embeds(gold.weights["text_embed.weight"], gold.prompt_embeds, gold.meta["token_len"])
    assert torch.equal(got, gold.t["in/prompt"])


def test_decoder_call_matches_reference(gold):
    """One DiffusionMLP.forward(x, t, z, pred_ids) captured inside the reference's denoise loop."""
    m = gold.meta
    out = O.diffusion_mlp(gold.weights, "image_decoder.", m["decoder_depth"], gold.t["dec/x"], gold.t["dec/t"],
                          gold.t["dec/z"], gold.t["dec/pred_ids"], m["patch"])
    ref = gold.t["dec/out"]
    assert (out - ref).abs().max() <= 1e-5 * ref.abs().max()


def _run(gold, dtype=torch.float32, replay=False):
    m, trace = gold.meta, {}
    kw = dict(u_dist=gold.t["in/u_dist"], noises=list(gold.t["in/noises"])) if replay else dict(
        generator=torch.Generator().manual_seed(m["sample_seed"]))
    x = O.generate(gold.weights, gold.oracle_config(), gold.t["in/prompt"], gold.t["in/num_preds"].numpy(),
                   num_diffusion_steps=m["S"], guidance_scale=m["guidance"], dtype=dtype, trace=trace, **kw)
    return x, trace


def test_generate_matches_reference_f32(gold):
    """End to end from the same seed: the oracle must consume the generator exactly like the reference."""
    x, trace = _run(gold)
    ref = gold.t["out/x"]
    assert torch.equal(trace["order"], gold.t["out/order"])
    scale = ref.abs().max()
    assert (trace["c"] - gold.t["out/c"]).abs().max() <= 1e-5 * gold.t["out/c"].abs().max()
    assert (trace["z"][0] - gold.t["out/z_first"]).abs().max() <= 1e-5 * gold.t["out/z_first"].abs().max()
    assert (trace["z"][-1] - gold.t["out/z_last"]).abs().max() <= 2e-5 * gold.t["out/z_last"].abs().max()
    assert x.shape == ref.shape
    assert (x - ref).abs().max() <= 1e-5 * scale, f"max rel err {((x - ref).abs().max() / scale).item():.3e}"


def test_generate_f64_replay_close_to_reference(gold):
    """float64 oracle on the replayed noise: bounds the f32 rounding of the reference itself (SURVEY: ~2e-6)."""
    x, _ = _run(gold, torch.float64, replay=True)
    ref = gold.t["out/x"].double()
    assert (x - ref).abs().max() <= 5e-5 * ref.abs().max()


def test_guidance_trunc_and_renorm_match_reference(gold):
    """guidance_trunc=450 (timestep units) + guidance_renorm=0.3, run by the reference's own GuidanceScaler / denoise loop."""
    m = gold.meta
    x = O.generate(gold.weights, gold.oracle_config(), gold.t["in/prompt"], gold.t["in/num_preds"].numpy(),
                   num_diffusion_steps=m["S"], guidance_scale=m["guidance"], guidance_trunc=450.0, guidance_renorm=0.3,
                   generator=torch.Generator().manual_seed(m["sample_seed"]))
    ref = gold.t["out/x_trunc450_renorm03"]
    assert (x - ref).abs().max() <= 1e-5 * ref.abs().max()
    assert (ref - gold.t["out/x"]).abs().max() > 1e-3 * ref.abs().max()  # the options really change the result


@pytest.mark.parametrize("key,with_prompt", [("out/x_rows_then_text", True), ("out/x_rows_only", False)])
def test_caller_supplied_condition_rows_match_reference(key, with_prompt):
    """inputs["c"] given by the caller (transformer_3d.py:66-77), run by the reference on the model of tiny_rope.npz
    (tests/golden/make_golden_c_rows.py): rows | TextEmbed(prompt) as the prefix, and the rows alone when there is no prompt."""
    from golden_util import c_rows_case

    gold, rows, outs = c_rows_case()
    m = gold.meta
    x = O.generate(gold.weights, gold.oracle_config(), gold.t["in/prompt"] if with_prompt else None, gold.t["in/num_preds"].numpy(),
                   num_diffusion_steps=m["S"], guidance_scale=m["guidance"], generator=torch.Generator().manual_seed(m["sample_seed"]),
                   c_pre=rows)
    ref = outs[key]
    assert x.shape == ref.shape and (x - ref).abs().max() <= 1e-5 * ref.abs().max()
    assert (ref - gold.t["out/x"]).abs().max() > 1e-3 * ref.abs().max()  # the rows change the result


def test_ddpm_plan_basics():
    plan = O.ddpm_plan(10, num_train_timesteps=100)
    assert [p[0] for p in plan] == list(range(90, -1, -10))
    assert plan[-1][5] == 0.0 and all(p[5] > 0 for p in plan[:-1])  # no noise at t = 0
    t, kx, kv, c0, cx, sigma = plan[-1]
    assert abs(cx) < 1e-6 and abs(c0 - 1.0) < 1e-6  # last step returns the predicted x0


# t: (kx, kv) of epsilon-prediction, c0, cx, sigma(fixed_small), sigma(fixed_large) - see the test below
DDPM_KNOWN_ANSWER = {
    2: (1.4085904245475278, -0.9920317455237933, 0.8415738934319075, 0.15087328172475567, 0.29784169859063525, 0.66332495807108),
    0: (1.0540925533894598, -0.3333333333333333, 1.0, 0.0, 0.0, 0.0),
}


def test_ddpm_known_answer_derived_by_hand_from_the_source_text():
    """a20 (parity unpinned by execution: the reference's DDPMScheduler imports diffusers): the coefficients of a 2-step chain over
    4 training timesteps with linear betas, worked out BY HAND from the formulas in scheduling_ddpm.py:143-146 (betas = linspace
    (0.1, 0.4, 4) -> alphas_cumprod = 0.9, 0.72, 0.504, 0.3024), :196-199 (leading spacing: timesteps 2, 0), :319-325 (previous
    timestep = t - 4 // 2), :268-285 (alpha_t / alpha_prev, predicted x0), :289-294 (mu coefficients), :211-222,297-305
    (fixed_small: variance = (1 - a_prev) / (1 - a_t) * beta_t, its square root scales the noise, none at t = 0; fixed_large: beta_t).
    The literals below are those numbers (DDPM_KNOWN_ANSWER; tests/test_mirror_cpu.py holds the drop-in's scheduler class and the HIP
    engine's per-step plan to the same literals)."""
    want = DDPM_KNOWN_ANSWER
    kw = dict(num_train_timesteps=4, beta_start=0.1, beta_end=0.4)
    for vt, col in (("fixed_small", 4), ("fixed_large", 5)):
        plan = O.ddpm_plan(2, variance_type=vt, **kw)
        assert [p[0] for p in plan] == [2, 0]
        for t, kx, kv, c0, cx, sigma in plan:
            w = want[t]
            for got, exp in ((kx, w[0]), (kv, w[1]), (c0, w[2]), (cx, w[3]), (sigma, w[col])):
                assert abs(got - exp) <= 2e-6, (vt, t, got, exp)
    vp = O.ddpm_plan(2, prediction_type="v_prediction", **kw)[0]  # :277-278: x0 = sqrt(a_t) x - sqrt(1 - a_t) v
    assert abs(vp[1] - 0.7099295739719539) <= 2e-6 and abs(vp[2] + 0.7042726744663603) <= 2e-6


# ---------------------------------------------------------------------------------------------
# multi-frame generation (KV-cached conditioning encoder, frame mixer, motion tokens) and 3-pass guidance:
# the oracle against runs of the reference's own generate_video (tests/golden/make_golden_video.py)
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module", params=VIDEO_CASES)
def vgold(request):
    return Golden(request.param)


def _vrun(vgold, **kw):
    m = vgold.meta
    args = dict(num_diffusion_steps=m["S"], guidance_scale=m["guidance"], max_latent_length=m["T"], motion_flow=[m["flow"]] * m["B"],
                generator=torch.Generator().manual_seed(m["sample_seed"]))
    args.update(kw)
    return O.generate(vgold.weights, vgold.oracle_config(), vgold.t["in/prompt"], vgold.t["in/num_preds"].numpy(), **args)


def test_video_generate_matches_reference(vgold):
    trace = {}
    x = _vrun(vgold, trace=trace)
    ref = vgold.t["out/x"]
    assert x.shape == ref.shape and x.shape[2] == vgold.meta["T"]
    assert torch.equal(trace["order"], vgold.t["out/order"])
    cf = torch.stack(trace["c_frames"])
    assert (cf - vgold.t["out/c_frames"]).abs().max() <= 2e-5 * vgold.t["out/c_frames"].abs().max()
    assert (x - ref).abs().max() <= 2e-5 * ref.abs().max()
    # replaying the recorded draws gives the same frames (RNG contract: one uniform, one normal per AR step per frame)
    y = _vrun(vgold, generator=None, u_dist=vgold.t["in/u_dist"], noises=list(vgold.t["in/noises"]))
    assert (y - ref).abs().max() <= 2e-5 * ref.abs().max()


@pytest.mark.parametrize("key,kw", [("out/x_image_guidance", dict(image_guidance_scale=1.5)),
                                    ("out/x_spatiotemporal_guidance", dict(spatiotemporal_guidance_scale=0.75)),
                                    ("out/x_image_guidance_renorm", dict(image_guidance_scale=1.5, guidance_renorm=0.4, guidance_trunc=300.0))])
def test_three_pass_guidance_matches_reference(vgold, key, kw):
    x = _vrun(vgold, **kw)
    ref = vgold.t[key]
    assert (x - ref).abs().max() <= 2e-5 * ref.abs().max()


def test_prefilled_first_frame_matches_reference(vgold):
    first = vgold.t["out/x"][:, :, 0]
    x = _vrun(vgold, latents=[first])
    ref = vgold.t["out/x_prefilled"]
    assert x.shape == ref.shape and torch.equal(x[:, :, 0], first)
    assert (x - ref).abs().max() <= 2e-5 * ref.abs().max()


def test_stored_schedule_oracle_outputs_are_well_formed():
    """tests/golden/schedule_oracle_<case>.npz (made by make_golden_schedule_oracle.py: the oracle's output for the full-schedule parity
    cases of test_gpu_parity_full.py): plain arrays (loadable without pickle), finite latents of the case's shape, the seeds the GPU test's
    fixture uses."""
    import glob
    import os

    import numpy as np

    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "schedule_oracle_*.npz")))
    assert len(files) >= 2
    for path in files:
        z = np.load(path, allow_pickle=False)
        width, heads, H, W, B, K, S = (int(v) for v in z["params"])
        assert z["ref"].shape == (B, 3, 1, H, W) and z["ref"].dtype == np.float32 and np.isfinite(z["ref"]).all()
        assert [int(v) for v in z["seeds"]] == [0, 4321, 29] and (K, S) == (64, 25) and width % heads == 0
        assert 1.0 < float(np.abs(z["ref"]).max()) < 10.0  # latents of a flow-matching sampler started from N(0, 1)
