"""Learning-rate schedule values from the reference's own classes (build container only):

    python tests/golden/make_golden_lr.py

Imports /root/reference/diffnext/engine/lr_scheduler.py by path (pure Python) and records 45 steps of get_lr() / step() for
ConstantLR, CosineLR (with and without decay_step / warm-up) and MultiStepLR -> tests/golden/lr_schedules.json.
"""
import importlib.util
import json
import os
import sys

sys.dont_write_bytecode = True
spec = importlib.util.spec_from_file_location("ref_lr", "/root/reference/diffnext/engine/lr_scheduler.py")
R = importlib.util.module_from_spec(spec)
spec.loader.exec_module(R)

CASES = [("ConstantLR", dict(lr_max=1e-3, lr_min=1e-5, warmup_steps=5, warmup_factor=0.01)),
         ("CosineLR", dict(lr_max=1e-3, max_steps=40, lr_min=1e-5, decay_step=3, warmup_steps=4)),
         ("CosineLR", dict(lr_max=2.0, max_steps=17)),
         ("MultiStepLR", dict(lr_max=1.0, decay_steps=[3, 9, 20], decay_gamma=0.5, warmup_steps=2))]
out = []
for name, kw in CASES:
    s = getattr(R, name)(**kw)
    vals = []
    for _ in range(45):
        vals.append(s.get_lr())
        s.step()
    out.append({"schedule": name, "kwargs": kw, "lr": vals})
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lr_schedules.json")
json.dump(out, open(path, "w"), indent=1)
print("->", path)
