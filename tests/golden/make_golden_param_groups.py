"""Optimizer parameter groups from the reference's own `get_param_groups` (build container only):

    python tests/golden/make_golden_param_groups.py

Imports /root/reference/diffnext/engine/engine_utils.py by path and applies it to the reference-built toy generator of
make_golden.py (after NOVATrainT2IPipeline.configure_model's freezes, restated there because the class imports diffusers),
with `lr_scale` / `no_weight_decay` attributes set on two parameters -> tests/golden/param_groups.json (parameter NAMES
per group + the group attributes).
"""
import importlib.util
import json
import os
import sys

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

from make_golden import build_model  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_engine_utils", "/root/reference/diffnext/engine/engine_utils.py")
U = importlib.util.module_from_spec(spec)
spec.loader.exec_module(U)

torch.manual_seed(0)
m = build_model(128, 2, (2, 2, 2), (8, 16), 3, 1, 64, 8, True)
# pipeline_train_t2i.py:65-68 (RESTATED: frozen modules of text-to-sample training), through the reference's freeze_module
U.freeze_module(m.text_embed.norm)
U.freeze_module(m.video_pos_embed)
U.freeze_module(m.video_encoder.patch_embed)
m.mask_embed.mask_token.lr_scale = 0.5
m.image_decoder.head.bias.no_weight_decay = True
names = {id(p): n for n, p in m.named_parameters()}
groups = [{"attrs": {k: v for k, v in g.items() if k != "params"}, "params": [names[id(p)] for p in g["params"]]}
          for g in U.get_param_groups(m)]
out = {"groups": groups, "count_params_M": U.count_params(m), "count_all_M": U.count_params(m, trainable=False)}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "param_groups.json")
json.dump(out, open(path, "w"), indent=1)
print("->", path, [(g["attrs"], len(g["params"])) for g in groups], out["count_params_M"])
