"""Expected outputs of the ORACLE (oracle/nova_oracle.py, the CPU restatement pinned by the other fixtures of this directory) for the
full-schedule parity cases of tests/test_gpu_parity_full.py whose oracle run is too long for a GPU-box test call.

    python tests/golden/make_golden_schedule_oracle.py headline_full_d48w1024_2048pts_K64S25

builds the case exactly as the test's fixture does (random-init architecture under torch.manual_seed(0) = bench.build_pipeline, prompts
from seed 4321, host generator seeded 29), runs `oracle.generate` and stores its output with the case's parameters in
tests/golden/schedule_oracle_<case>.npz. The test then compares the HIP path with this stored result instead of re-running the oracle.
The file holds data only (the latents, float32); nothing of the reference is involved.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "nova_pointcloud_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import bench  # noqa: E402
from oracle import nova_oracle as O  # noqa: E402

CASES = {  # (width, heads, latent H, W, batch, AR steps, diffusion steps)
    "headline_full_d48w1024_2048pts_K64S25": (1024, 16, 32, 64, 1, 64, 25),
    "config1_full_d48w768_1024pts_K64S25": (768, 12, 32, 32, 1, 64, 25),
}


def main(name):
    width, heads, H, W, B, K, S = CASES[name]
    torch.set_num_threads(bench.host_cores())
    pipe = bench.build_pipeline(width, heads, H, W, torch.float32, torch.device("cpu"))
    sd = {k: v.detach().clone() for k, v in pipe.transformer.state_dict().items()}
    prompts = bench.synthetic_prompts(B, "cpu", torch.float32, seed=4321)
    N = H * W
    sched = [int(v) for v in O.cosine_schedule(N, K) if v > 0]
    cfg = O.make_config(3, (H, W), 1, width, heads, 16, 32, 6, 256, rotary=True)
    prompt = O.encode_prompt_embeds(sd["text_embed.weight"], prompts, 256)
    t0 = time.time()
    with torch.no_grad():
        ref = O.generate(sd, cfg, prompt, sched, num_diffusion_steps=S, guidance_scale=5.0, generator=torch.Generator().manual_seed(29))
    dt = time.time() - t0
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"schedule_oracle_{name}.npz")
    np.savez_compressed(out, ref=ref.float().numpy(), params=np.array([width, heads, H, W, B, K, S], dtype=np.int64),
                        seeds=np.array([0, 4321, 29], dtype=np.int64), oracle_seconds=np.array([dt]), torch_version=np.array(torch.__version__))
    print(f"{name}: oracle {dt:.0f} s on {torch.get_num_threads()} threads, output {tuple(ref.shape)}, |x| max {ref.abs().max().item():.4f} -> {out}", flush=True)


if __name__ == "__main__":
    main(sys.argv[1])
