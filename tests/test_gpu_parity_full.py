"""End-to-end parity of the HIP generation path with the oracle at the BASELINE architectures.

The golden fixtures (tests/golden) pin the oracle on 2-block, D=128 models. What they cannot show is
error growth through the real depth: 16 conditioning blocks + 32 masked-AR blocks + 6 diffusion blocks
at D = 768 / 1024 / 1536. Here `NOVAPipeline` runs the real NOVA-d48 architectures (random-init under
`torch.manual_seed(0)`, the bench's weights) on the GPU and `oracle.generate` runs the same weights,
prompts and host generator on the CPU, at reduced AR / diffusion step counts so the oracle finishes in
seconds (SURVEY §8d parity chain (ii)/(iii); BASELINE.json configs[1], [2], [4]).

  f32 mode   (exact-f32 MFMA): max|d| / max|x| <= 1e-3 on the point coordinates (north_star).
  bf16 mode  (throughput mode): generation order and noise injected (the reference draws them in the
             activation dtype, so a bf16 seed is not comparable), rms-relative error reported and
             bounded; the reference's own CPU-bf16 level on the toy model is ~2e-2.
Measured errors are printed (`pytest -s`) and recorded in DESIGN.md §2.
"""
import os
import sys

import pytest
import torch

from oracle import nova_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nova_pointcloud_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()


def rms_rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


# (width, heads, latent H, W, batch, AR steps, diffusion steps, bf16 rms-rel bound)
ARCHS = {
    # measured on MI355X (profiles/r02_parity_full.log): f32 3.5e-5 / 1.7e-5 / 2.2e-5 max-rel; bf16 2.1e-2 / 1.8e-2 / 1.9e-2 rms-rel
    "d48w1024_2048pts": (1024, 16, 32, 64, 2, 3, 2, 4e-2),  # configs[2] / [3]: the headline architecture
    "d48w768_1024pts": (768, 12, 32, 32, 2, 3, 2, 4e-2),    # configs[1]
    "d48w1536_2048pts": (1536, 16, 32, 64, 1, 2, 2, 4e-2),  # configs[4] architecture (head_dim 96) in f32 / bf16
    # depth in SCHEDULE (round 3): the AR loop feeds its own output back K times and the Euler loop S times
    "config0_d48w768_256pts_K4S4": (768, 12, 16, 16, 1, 4, 4, 4e-2),     # BASELINE configs[0] IN FULL (256 points, 4 x 4 steps, batch 1)
    "sched_d48w768_1024pts_K16S8": (768, 12, 32, 32, 1, 16, 8, 8e-2),    # 16 AR x 8 diffusion steps: error growth over the schedule
}
# Full-schedule cases: 64 AR x 25 diffusion steps, batch 1. The oracle run for them takes minutes of host time (thousands of small
# operations at batch 1; the headline one does not finish inside a GPU-box call), so its OUTPUT is a committed fixture,
# tests/golden/schedule_oracle_<case>.npz, made by tests/golden/make_golden_schedule_oracle.py with exactly the construction of the
# fixture below (weights from torch.manual_seed(0), prompts from seed 4321, host generator 29). The case is in the suite when its file is.
# NOVA_PARITY_FULL_SCHEDULE=1 additionally re-runs the oracle live for configs[1] and checks the stored output against it.
FULL_SCHEDULE = {
    "config1_full_d48w768_1024pts_K64S25": (768, 12, 32, 32, 1, 64, 25, 8e-2),       # BASELINE configs[1]'s architecture and schedule
    "headline_full_d48w1024_2048pts_K64S25": (1024, 16, 32, 64, 1, 64, 25, 8e-2),    # the headline architecture and schedule
}
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def stored_oracle(name):
    path = os.path.join(GOLDEN_DIR, f"schedule_oracle_{name}.npz")
    return path if os.path.exists(path) else None


for _name, _case in FULL_SCHEDULE.items():
    if stored_oracle(_name) or (_name.startswith("config1_") and os.environ.get("NOVA_PARITY_FULL_SCHEDULE") == "1"):
        ARCHS[_name] = _case
SCHEDULE_CASES = tuple(k for k in ARCHS if k.startswith(("config0_", "sched_", "config1_full_", "headline_full_")))


@pytest.fixture(scope="module", params=sorted(ARCHS))
def case(request):
    """Builds the architecture once (CPU f32 master copy), runs the oracle once, and hands both to the tests."""
    import bench

    width, heads, H, W, B, K, S, bf16_bound = ARCHS[request.param]
    threads = torch.get_num_threads()
    torch.set_num_threads(max(threads, bench.host_cores()))
    pipe = bench.build_pipeline(width, heads, H, W, torch.float32, torch.device("cpu"))
    sd = {k: v.detach().clone() for k, v in pipe.transformer.state_dict().items()}
    prompts = bench.synthetic_prompts(B, "cpu", torch.float32, seed=4321)
    N = H * W
    # the draws a host generator seeded with 29 yields in the path's order (embeddings.py:265, transformer_3d.py:131): one
    # uniform [B,N,1], then one normal [B,C,H,W] per AR step. Injecting them must reproduce the seeded run, so ONE oracle
    # run serves the seeded f32 test and the injected bf16 test.
    g = torch.Generator().manual_seed(29)
    u_dist = torch.empty(B, N, 1).uniform_(generator=g)
    sched = [int(v) for v in O.cosine_schedule(N, K) if v > 0]
    noises = [torch.empty(B, 3, H, W).normal_(generator=g) for _ in sched]
    cfg = O.make_config(3, (H, W), 1, width, heads, 16, 32, 6, 256, rotary=True)
    prompt = O.encode_prompt_embeds(sd["text_embed.weight"], prompts, 256)
    stored = stored_oracle(request.param)
    live = stored is None or os.environ.get("NOVA_PARITY_FULL_SCHEDULE") == "1" and request.param.startswith("config1_")
    ref_stored = None
    if stored:
        import numpy as np

        z = np.load(stored)
        assert [int(v) for v in z["params"]] == [width, heads, H, W, B, K, S] and [int(v) for v in z["seeds"]] == [0, 4321, 29]
        ref_stored = torch.from_numpy(z["ref"])
    if not live:
        torch.set_num_threads(threads)
        return dict(name=request.param, pipe=pipe, prompts=prompts, K=K, S=S, order=u_dist.argsort(dim=1)[..., 0], noises=noises,
                    ref=ref_stored, bf16_bound=bf16_bound, shape=(B, 3, 1, H, W))
    # the oracle run is silent host work (minutes at the full schedule): a line per minute on stderr keeps a watchdog that
    # kills silent commands from mistaking it for a hang
    import threading
    import time
    done = threading.Event()

    def heartbeat(t0=time.time()):
        while not done.wait(60.0):
            print(f"[parity-full] {request.param}: oracle running on the host, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)

    beat = threading.Thread(target=heartbeat, daemon=True)
    beat.start()
    try:
        with torch.no_grad():
            ref = O.generate(sd, cfg, prompt, sched, num_diffusion_steps=S, guidance_scale=5.0, generator=torch.Generator().manual_seed(29))
    finally:
        done.set()
        beat.join()
    torch.set_num_threads(threads)
    if ref_stored is not None:  # the stored output against the oracle re-run on this host: the same arithmetic on another machine
        drift = rel(ref_stored, ref)
        print(f"\n[parity-full] {request.param}: stored oracle output vs live oracle run: max rel {drift:.3e}")
        assert drift < 1e-4, drift
    return dict(name=request.param, pipe=pipe, prompts=prompts, K=K, S=S, order=u_dist.argsort(dim=1)[..., 0], noises=noises,
                ref=ref, bf16_bound=bf16_bound, shape=(B, 3, 1, H, W))


def run(case, dtype, **kw):
    """One pipeline call on a GPU copy of the model in `dtype` (the CPU f32 master copy stays untouched)."""
    import copy

    from diffnext.pipelines import NOVAPipeline
    from diffnext.schedulers import FlowMatchEulerDiscreteScheduler

    model = copy.deepcopy(case["pipe"].transformer).to(device="cuda", dtype=dtype).eval()
    pipe = NOVAPipeline(transformer=model, scheduler=FlowMatchEulerDiscreteScheduler(num_train_timesteps=1000, shift=1.0))
    out = pipe(prompt_embeds=[p.to("cuda", dtype) for p in case["prompts"]], num_inference_steps=case["K"],
               num_diffusion_steps=case["S"], guidance_scale=5, output_type="latent", disable_progress_bar=True, **kw).frames
    torch.cuda.synchronize()
    return out.float().cpu()


def test_f32_from_seed_matches_oracle_at_full_depth(case, hip):
    """Same prompts + same host generator seed: HIP f32 path vs the oracle, 1e-3 relative on the coordinates."""
    x = run(case, torch.float32, generator=torch.Generator().manual_seed(29))
    assert tuple(x.shape) == case["shape"] and torch.isfinite(x).all()
    err, rms = rel(x, case["ref"]), rms_rel(x, case["ref"])
    print(f"\n[parity-full] {case['name']} f32 from seed: max rel {err:.3e}, rms rel {rms:.3e}")
    assert err < 1e-3, err
    assert err < 2e-4, f"f32 MFMA path measured 2-4e-5 at full depth, got {err:.3e}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_16bit_injected_draws_close_to_oracle_at_full_depth(case, hip, dtype):
    """Throughput modes (bf16 / f16 storage, f32 accumulate): weights rounded to the storage type on the GPU side only - the
    error reported is the whole 16-bit effect against the f32 reference, as north_star's 'bf16 vs reference CPU path';
    float16 (3 more mantissa bits, the reference callers' default: scripts/app_nova_t2i.py:36) is held to a quarter of it."""
    order, noises = case["order"], case["noises"]
    x = run(case, dtype, pred_order=order, noise_fn=lambda i: noises[i])
    assert torch.isfinite(x).all()
    err, mx = rms_rel(x, case["ref"]), rel(x, case["ref"])
    print(f"\n[parity-full] {case['name']} {str(dtype).split('.')[-1]} injected: rms rel {err:.3e}, max rel {mx:.3e}")
    assert err < case["bf16_bound"] * (1.0 if dtype == torch.bfloat16 else 0.25), err


def test_fp8_gemm_mode_against_bf16_and_oracle_at_full_depth(case, hip):
    """BASELINE configs[4]: the encoder's QKV / fc1 / fc2 GEMMs on the block-scaled fp8 MFMA. The
    reference has no fp8 path, so the result is compared with this build's bf16 output under the same injected order and
    noise (rms-relative, reported), and with the f32 oracle for scale."""
    if case["name"] in SCHEDULE_CASES:
        pytest.skip("fp8 mode is measured at the three architecture cases")
    order, noises = case["order"], case["noises"]
    kw = dict(pred_order=order, noise_fn=lambda i: noises[i])
    x16 = run(case, torch.bfloat16, **kw)
    x8 = run(case, torch.bfloat16, gemm_dtype="fp8", **kw)
    assert torch.isfinite(x8).all()
    e_bf16, e_ref = rms_rel(x8, x16), rms_rel(x8, case["ref"])
    print(f"\n[parity-full] {case['name']} fp8 GEMMs: rms rel vs bf16 {e_bf16:.3e}, vs f32 oracle {e_ref:.3e}")
    assert e_bf16 < 0.092, e_bf16  # measured 7.0-7.7e-2 over rounds 2-4 + 20 %


def test_fp8_gemm_mode_is_a_function_of_the_call_alone(case, hip):
    """The fp8 GEMM mode's delayed scaling (per layer: the e4m3 scale of the MLP hidden rows follows the largest |GELU| the
    previous AR step of the SAME call saw) keeps no state between calls: call 1 == call 2 on one pipeline == the first call of a
    fresh pipeline, bit for bit, at the real depth. The scales are a statistic of a lane's rows, so the LANE count (a scheduling
    choice) does show in this mode; with `fp8_row_scaled_hidden=True` (each hidden row quantised with its own scale, one extra
    pass per block) no statistic crosses a row and 1 lane == 2 lanes bit for bit."""
    import copy

    from diffnext.pipelines import NOVAPipeline
    from diffnext.schedulers import FlowMatchEulerDiscreteScheduler

    if case["name"] != "d48w1024_2048pts":
        pytest.skip("one architecture with a batch of 2 is enough")
    order, noises = case["order"], case["noises"]

    def fresh():
        model = copy.deepcopy(case["pipe"].transformer).to(device="cuda", dtype=torch.bfloat16).eval()
        return NOVAPipeline(transformer=model, scheduler=FlowMatchEulerDiscreteScheduler(num_train_timesteps=1000, shift=1.0))

    def call(pipe, **kw):
        out = pipe(prompt_embeds=[p.to("cuda", torch.bfloat16) for p in case["prompts"]], num_inference_steps=case["K"],
                   num_diffusion_steps=case["S"], guidance_scale=5, output_type="latent", disable_progress_bar=True,
                   pred_order=order, noise_fn=lambda i: noises[i], gemm_dtype="fp8", **kw).frames
        torch.cuda.synchronize()
        return out

    pipe = fresh()
    first = call(pipe)
    bf16 = pipe(prompt_embeds=[p.to("cuda", torch.bfloat16) for p in case["prompts"]], num_inference_steps=case["K"],
                num_diffusion_steps=case["S"], guidance_scale=5, output_type="latent", disable_progress_bar=True,
                pred_order=order, noise_fn=lambda i: noises[i]).frames  # a bf16 call in between must not matter either
    second = call(pipe)
    other = call(fresh())
    assert torch.isfinite(first.float()).all() and not torch.equal(first, bf16)
    assert torch.equal(first, second) and torch.equal(first, other)
    r1, r2 = call(pipe, fp8_row_scaled_hidden=True, lanes=1), call(pipe, fp8_row_scaled_hidden=True, lanes=2)
    assert torch.equal(r1, r2)
    assert rms_rel(r1.float().cpu(), case["ref"]) < 0.092


def test_benchmarked_workload_batch32_is_its_batch1_samples_and_the_oracle_case(hip):
    """The workload `bench.py` times - BASELINE configs[2]: NOVA-d48w1024, 2048 points, batch 32, 64 AR x 25 diffusion steps, bf16,
    a DEVICE generator, two half-batch lanes (256-tile GEMMs at M = 163840, the hoisted AdaLN buffer, graph-replayed denoising
    loops) - run inside the suite and tied to the oracle (transformer_3d.py:115-133):
      (a) the bench's own call (its prompts, its generator seed) gives, for samples 0 / 15 / 16 / 31 (first and last of either
          lane), bit for bit the points the same prompt gives ALONE at batch 1 under the same draws (injected from a second
          generator with the same seed: uniform [B, N, 1] first, then one normal [B, C, H, W] per AR step);
      (b) with the stored oracle case's prompt and draws in slot 0 (tests/golden/schedule_oracle_headline_full_*.npz: weights from
          seed 0 = the bench's, prompt from seed 4321, host generator 29), sample 0 of the batch of 32 is bit for bit that case
          run alone, within the bf16 bound of the oracle's output, and the other 31 samples are unchanged from (a)."""
    import numpy as np

    import bench

    name = "headline_full_d48w1024_2048pts_K64S25"
    path = stored_oracle(name)
    if path is None:
        pytest.skip("stored oracle output of the headline case is not in tests/golden")
    width, heads, H, W, B = bench.WORKLOADS["d48w1024_2048pts_b32"]
    N, K, S = H * W, 64, 25
    dev = torch.device("cuda")
    z = np.load(path)
    assert [int(v) for v in z["params"]] == [width, heads, H, W, 1, K, S] and [int(v) for v in z["seeds"]] == [0, 4321, 29]
    ref = torch.from_numpy(z["ref"])  # [1, 3, 1, H, W]
    pipe = bench.build_pipeline(width, heads, H, W, torch.bfloat16, dev)  # weights: torch.manual_seed(0), as the stored case's
    prompts = bench.synthetic_prompts(B, dev, torch.bfloat16, seed=1234)  # bench.py's prompts

    def call(prm, **kw):
        out = pipe(prompt_embeds=prm, num_inference_steps=K, num_diffusion_steps=S, guidance_scale=5, output_type="latent",
                   disable_progress_bar=True, **kw).frames
        return out

    # (a) bench.py's call
    got = call(prompts, generator=torch.Generator(device=dev).manual_seed(0))
    g = torch.Generator(device=dev).manual_seed(0)
    order = torch.empty(B, N, 1, device=dev).uniform_(generator=g).argsort(dim=1)[..., 0]
    n_steps = len([v for v in O.cosine_schedule(N, K) if v > 0])
    noises = [torch.empty(B, 3, H, W, device=dev).normal_(generator=g).clone() for _ in range(n_steps)]
    torch.cuda.synchronize()
    assert got.shape == (B, 3, 1, H, W) and torch.isfinite(got.float()).all()
    for i in (0, 15, 16, 31):
        alone = call([prompts[i]], pred_order=order[i:i + 1], noise_fn=lambda s, i=i: noises[s][i:i + 1])
        assert torch.equal(alone, got[i:i + 1]), (i, (alone.float() - got[i:i + 1].float()).abs().max().item())
    # (b) the oracle case in slot 0
    hg = torch.Generator().manual_seed(29)
    u0 = torch.empty(1, N, 1).uniform_(generator=hg)
    n0 = [torch.empty(1, 3, H, W).normal_(generator=hg) for _ in range(n_steps)]
    p0 = bench.synthetic_prompts(1, dev, torch.bfloat16, seed=4321)
    order_b = torch.cat([u0.argsort(dim=1)[..., 0].to(dev), order[1:]])
    noises_b = [torch.cat([n0[s].to(dev), noises[s][1:]]) for s in range(n_steps)]
    mixed = call(p0 + prompts[1:], pred_order=order_b, noise_fn=lambda s: noises_b[s])
    alone0 = call(p0, pred_order=order_b[:1], noise_fn=lambda s: noises_b[s][:1])
    torch.cuda.synchronize()
    assert torch.equal(mixed[1:], got[1:])
    assert torch.equal(mixed[:1], alone0)
    err = rms_rel(mixed[:1].float(), ref)
    print(f"\n[parity-full] batch-32 headline workload, sample 0 = stored oracle case: bf16 rms rel {err:.3e}")
    assert err < FULL_SCHEDULE[name][7], err
