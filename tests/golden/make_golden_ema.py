"""EMA weights from the reference's own ModelEMA (build container only; diffnext/engine/model_ema.py imported by path):

    python tests/golden/make_golden_ema.py   ->   tests/golden/model_ema.json
"""
import importlib.util
import json
import os
import sys

import torch

sys.dont_write_bytecode = True
spec = importlib.util.spec_from_file_location("ref_ema", "/root/reference/diffnext/engine/model_ema.py")
R = importlib.util.module_from_spec(spec)
spec.loader.exec_module(R)


def scenario(ModelEMA):
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.LayerNorm(3)).to(torch.bfloat16)
    net[1].bias.requires_grad = False  # frozen parameters are not averaged
    ema = ModelEMA(net, decay=0.9, update_every=2)
    g = torch.Generator().manual_seed(4)
    for _ in range(3):
        with torch.no_grad():
            for p in net.parameters():
                p.add_(torch.randn(p.shape, generator=g).to(p.dtype) * 0.1)
        ema.update(net)
    return {k: v.float().flatten().tolist() for k, v in ema.model.state_dict().items()}, [str(v.dtype) for v in ema.model.state_dict().values()]


if __name__ == "__main__":
    vals, dtypes = scenario(R.ModelEMA)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_ema.json")
    json.dump({"values": vals, "dtypes": dtypes}, open(path, "w"), indent=1)
    print("->", path, dtypes)
