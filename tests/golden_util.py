"""Loader for tests/golden/*.npz (made by tests/golden/make_golden.py from the reference's modules)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["tiny_rope", "tiny_abspe"]
VIDEO_CASES = ["tiny_video_rope", "tiny_video_abspe"]  # tests/golden/make_golden_video.py: T > 1, mixer, motion tokens, 3-pass guidance


def bf16_bits_to_f32(a):
    return torch.from_numpy(a.astype(np.int16)).view(torch.bfloat16).float()


class Golden(object):
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.name = name
        self.weights = {k[2:]: bf16_bits_to_f32(z[k]) for k in z.files if k.startswith("w/")}
        self.meta = {k[5:]: z[k].item() for k in z.files if k.startswith("meta/")}
        self.t = {k: torch.from_numpy(z[k]) for k in z.files if k.split("/")[0] in ("in", "out", "dec", "train")}
        n = len([k for k in z.files if k.startswith("in/prompt_embeds/")])
        self.prompt_embeds = [self.t[f"in/prompt_embeds/{i}"] for i in range(n)]

    def oracle_config(self):
        from oracle import nova_oracle as O

        m = self.meta
        return O.make_config(m["image_dim"], (m["latent_h"], m["latent_w"]), m["patch"], m["D"], m["heads"],
                             m["video_depth"], m["image_depth"], m["decoder_depth"], m["token_len"], bool(m["rotary"]),
                             video_base_t=m.get("T", 1))


def c_rows_case():
    """tests/golden/tiny_rope_c_rows.npz (make_golden_c_rows.py: the reference run with a caller-supplied condition list on the model of
    tiny_rope.npz): (Golden("tiny_rope"), [rows_a, rows_b], {"out/x_rows_then_text": x, "out/x_rows_only": x})."""
    z = np.load(os.path.join(GOLDEN_DIR, "tiny_rope_c_rows.npz"), allow_pickle=False)
    rows = [torch.from_numpy(z[f"in/c_rows/{i}"]) for i in range(2)]
    return Golden("tiny_rope"), rows, {k: torch.from_numpy(z[k]) for k in z.files if k.startswith("out/")}
