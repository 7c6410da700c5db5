"""Golden vectors for the caller-supplied condition list inputs["c"] (transformer_3d.py:66-77), by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference, read-only):

    python tests/golden/make_golden_c_rows.py        -> tests/golden/tiny_rope_c_rows.npz

The model is the one of tiny_rope.npz (same constructors, same seeds; checked weight by weight against the stored file, so the new
fixture holds inputs and outputs only). Two reference runs of Transformer3DModel.forward from the fixture's sample seed:
  out/x_rows_then_text   inputs["c"] = [rows_a, rows_b] and the fixture's prompt: prefix = rows_a | rows_b | TextEmbed(prompt)
  out/x_rows_only        inputs["c"] = [rows_a, rows_b], no prompt: the prefix is the given rows alone (the class-conditional
                         pipeline's call form, pipeline_nova_c2i.py:88)
Rows are [2B, Lc, D] ([conditional ; unconditional] blocks, as `prompt` is), on the bf16 grid.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (imports the reference's modules from /root/reference)

if __name__ == "__main__":
    seed, D, heads, depths, latent_hw, token_dim, token_len, B, K, S = 0, 128, 2, (2, 2, 2), (8, 16), 64, 8, 2, 5, 3
    stored = np.load(os.path.join(HERE, "tiny_rope.npz"))
    torch.manual_seed(seed)
    m = G.build_model(D, heads, depths, latent_hw, 3, 1, token_dim, token_len, True)
    G.to_bf16_grid(m, seed + 1)
    for k, v in m.state_dict().items():
        assert np.array_equal(G.bf16_bits(v), stored["w/" + k]), k  # the very model of tiny_rope.npz
    prompt = torch.from_numpy(stored["in/prompt"])
    num_preds = stored["in/num_preds"]
    sample_seed = int(stored["meta/sample_seed"])
    g = torch.Generator().manual_seed(99)
    rows = [(torch.randn(2 * B, 3, D, generator=g) * 0.3).bfloat16().float(), (torch.randn(2 * B, 1, D, generator=g) * 0.3).bfloat16().float()]
    base = {"num_preds": num_preds, "guidance_scale": float(stored["meta/guidance"]), "batch_size": B, "num_diffusion_steps": S,
            "max_latent_length": 1, "tqdm1": False, "tqdm2": False, "guidance_trunc": 0, "guidance_renorm": 1,
            "image_guidance_scale": 0, "spatiotemporal_guidance_scale": 0}
    outs = {}
    for name, with_prompt in (("out/x_rows_then_text", True), ("out/x_rows_only", False)):
        inputs = dict(base, c=[r.clone() for r in rows], generator=torch.Generator().manual_seed(sample_seed))
        if with_prompt:
            inputs["prompt"] = prompt.clone()
        with torch.no_grad():
            outs[name] = m(inputs)["x"].numpy()
    assert not np.allclose(outs["out/x_rows_then_text"], stored["out/x"], atol=1e-3)
    arrays = {"in/c_rows/0": rows[0].numpy(), "in/c_rows/1": rows[1].numpy(), **outs}
    path = os.path.join(HERE, "tiny_rope_c_rows.npz")
    np.savez(path, **arrays)
    print({k: (v.shape, float(np.abs(v).max())) for k, v in arrays.items()}, "->", path, os.path.getsize(path))
