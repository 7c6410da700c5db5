"""CPU: the training side - learning-rate schedules, the Trainer loop with checkpoints and resume,
and one-process-per-device data parallelism (gloo, world_size 2): bucketed gradient averaging keeps the ranks identical."""
import math
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nova_pointcloud_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts")):
    if p not in sys.path:
        sys.path.insert(0, p)

from diffnext.engine.lr_scheduler import ConstantLR, CosineLR, MultiStepLR  # noqa: E402


def tiny_model(seed=0):
    from diffnext.models.transformers import transformer_nova as TN

    TN.VIDEO_ENCODERS.register("tr_vit_d1w64", TN._vit, depth=1, embed_dim=64, num_heads=2)
    TN.IMAGE_ENCODERS.register("tr_vit_d2w64", TN._vit, depth=2, embed_dim=64, num_heads=2)
    TN.IMAGE_DECODERS.register("tr_mlp_d1w64", TN._mlp, depth=1, embed_dim=64)
    torch.manual_seed(seed)
    return TN.NOVATransformer3DModel(image_dim=3, image_size=(64, 64), image_stride=16, text_token_dim=32, text_token_len=8,
                                     image_base_size=[4, 4], video_base_size=[1, 2, 2], rotary_pos_embed=True,
                                     arch=("tr_vit_d1w64", "tr_vit_d2w64", "tr_mlp_d1w64"))


def tiny_config(out_dir, steps=4, accum=1):
    return {"experiment": {"output_dir": str(out_dir), "log_every": 1, "save_every": 2},
            "model": {"name": "transformer", "loss_repeat": 2},
            "training": {"seed": 7, "max_train_steps": steps, "gradient_accumulation_steps": accum, "max_grad_norm": 1.0},
            "optimizer": {"target": "torch.optim.AdamW", "params": {"lr": 2e-3, "betas": [0.9, 0.95], "weight_decay": 0.02}},
            "lr_scheduler": {"target": "diffnext.engine.lr_scheduler.ConstantLR", "params": {"lr_max": 2e-3, "warmup_steps": 2}},
            "parallel": {"bucket_mb": 0.05}}  # small buckets: several all-reduces per step


def test_lr_schedules_follow_the_reference_formulas():
    c = ConstantLR(lr_max=1.0, warmup_steps=4, warmup_factor=0.1)
    got = []
    for _ in range(6):
        got.append(c.get_lr())
        c.step()
    want = [(a + (1 - a) * 0.1) for a in (0.25, 0.5, 0.75, 1.0)] + [1.0, 1.0]
    assert all(abs(g - w) < 1e-12 for g, w in zip(got, want))
    s = CosineLR(lr_max=1.0, max_steps=10, lr_min=0.1, warmup_steps=2, warmup_factor=0.5)
    vals = []
    for _ in range(10):
        vals.append(s.get_lr())
        s.step()
    assert abs(vals[0] - 0.75) < 1e-12 and abs(vals[1] - 1.0) < 1e-12 and abs(vals[2] - 1.0) < 1e-12  # t = 0: decay stays 1
    for t in range(1, 8):
        assert abs(vals[2 + t] - (0.1 + 0.9 * 0.5 * (1 + math.cos(math.pi * t / 8)))) < 1e-12
    m = MultiStepLR(lr_max=1.0, decay_steps=[2, 4], decay_gamma=0.1)
    seq = []
    for _ in range(6):
        seq.append(m.get_lr())
        m.step()
    assert [round(v, 6) for v in seq] == [1.0, 1.0, 0.1, 0.1, 0.01, 0.01]


def test_lr_schedules_match_reference_values():
    """tests/golden/lr_schedules.json: 45 steps of the reference's own ConstantLR / CosineLR / MultiStepLR objects."""
    import json

    from diffnext.engine import lr_scheduler as M

    for case in json.load(open(os.path.join(ROOT, "tests", "golden", "lr_schedules.json"))):
        sched = getattr(M, case["schedule"])(**case["kwargs"])
        for i, want in enumerate(case["lr"]):
            got = sched.get_lr()
            assert abs(got - want) <= 1e-15 * max(1.0, abs(want)), (case["schedule"], i, got, want)
            assert sched.get_lr() == got  # asking twice does not move the schedule
            sched.step()
        resumed = getattr(M, case["schedule"])(**case["kwargs"])
        resumed._step_count = 30  # the Trainer sets the step count on resume
        assert abs(resumed.get_lr() - case["lr"][30]) <= 1e-15 * max(1.0, abs(case["lr"][30]))


def test_param_groups_match_reference_function():
    """tests/golden/param_groups.json: the reference's own get_param_groups / freeze_module / count_params on its toy model."""
    import json

    from golden_util import Golden
    from test_mirror_cpu import build_from_golden

    from diffnext.engine import engine_utils as U

    want = json.load(open(os.path.join(ROOT, "tests", "golden", "param_groups.json")))
    m = build_from_golden(Golden("tiny_rope"))  # same architecture as the generator's model
    U.freeze_module(m.text_embed.norm), U.freeze_module(m.video_pos_embed), U.freeze_module(m.video_encoder.patch_embed)
    m.mask_embed.mask_token.lr_scale = 0.5
    m.image_decoder.head.bias.no_weight_decay = True
    names = {id(p): n for n, p in m.named_parameters()}
    got = [{"attrs": {k: v for k, v in g.items() if k != "params"}, "params": [names[id(p)] for p in g["params"]]}
           for g in U.get_param_groups(m)]
    assert got == want["groups"]
    assert abs(U.count_params(m) - want["count_params_M"]) < 1e-9 and abs(U.count_params(m, trainable=False) - want["count_all_M"]) < 1e-9


def test_model_ema_matches_reference_class():
    """tests/golden/model_ema.json (make_golden_ema.py): the reference's own ModelEMA after three updates of a bf16 toy network -
    f32 averages of the trainable parameters, frozen ones untouched. Same scenario on the mirror class."""
    import json

    from diffnext.engine.engine_utils import ModelEMA

    want = json.load(open(os.path.join(ROOT, "tests", "golden", "model_ema.json")))
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.LayerNorm(3)).to(torch.bfloat16)
    net[1].bias.requires_grad = False
    ema = ModelEMA(net, decay=0.9, update_every=2)
    g = torch.Generator().manual_seed(4)
    for _ in range(3):
        with torch.no_grad():
            for p in net.parameters():
                p.add_(torch.randn(p.shape, generator=g).to(p.dtype) * 0.1)
        ema.update(net)
    sd = ema.model.state_dict()
    assert [str(v.dtype) for v in sd.values()] == want["dtypes"]
    for k, v in sd.items():
        assert v.float().flatten().tolist() == want["values"][k], k
    assert ema.update_every == 2 and all(not p.requires_grad for p in ema.model.parameters())


class FixedBatch(object):
    def __init__(self, seed=3, B=4):
        from diffnext.engine.datasets import SyntheticPointClouds

        self.batch = SyntheticPointClouds(B, (4, 4), 32, seed=seed).next()

    def next(self):
        return [{"x": self.batch[0]["x"].clone(), "prompt": [p.clone() for p in self.batch[0]["prompt"]]}]


def test_trainer_loop_checkpoints_and_resume(tmp_path):
    from diffnext.engine.train_engine import Trainer
    from diffnext.models.transformers.transformer_nova import NOVATransformer3DModel
    from diffnext.schedulers import FlowMatchEulerDiscreteScheduler

    from diffnext.engine import engine_utils

    engine_utils.manual_seed(11)
    model = tiny_model()
    trainer = Trainer(tiny_config(tmp_path, steps=12), model, FixedBatch(), noise_scheduler=FlowMatchEulerDiscreteScheduler())
    frozen = [p for p in model.text_embed.norm.parameters()] + list(model.video_encoder.patch_embed.parameters())
    assert all(not p.requires_grad for p in frozen) and not model.text_embed.norm.training  # pipeline_train_t2i.py:65-68
    history = trainer.train_loop()
    assert [h["step"] for h in history] == list(range(12)) and trainer.global_step == 12
    first, last = sum(h["metrics"]["loss"] for h in history[:3]) / 3, sum(h["metrics"]["loss"] for h in history[-3:]) / 3
    assert math.isfinite(last) and last < first  # the same batch every step: the loss must go down
    ckpt = tmp_path / "checkpoints" / "checkpoint-12" / "transformer"
    assert (ckpt / "config.json").exists() and (tmp_path / "checkpoints" / "checkpoint-2").exists()
    again = NOVATransformer3DModel.from_pretrained(str(ckpt))
    for (k, a), (_, b) in zip(model.state_dict().items(), again.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(RuntimeError, match="Excepted a trainable model"):
        model.eval()
        model({"x": torch.zeros(1, 3, 4, 4), "prompt": [torch.zeros(2, 32)]})


def test_resumed_trainer_continues_the_saved_ema(tmp_path):
    """The EMA written beside a checkpoint is what a resumed run averages on from (reference engine/train_engine.py:52-56), not a
    fresh copy of the resumed raw weights; and a checkpoint stored the reference's way (diffusion_pytorch_model.bin) loads."""
    from diffnext.engine import engine_utils
    from diffnext.engine.train_engine import Trainer
    from diffnext.models.transformers.transformer_nova import NOVATransformer3DModel
    from diffnext.schedulers import FlowMatchEulerDiscreteScheduler

    engine_utils.manual_seed(11)
    cfg = tiny_config(tmp_path, steps=4)
    cfg["ema"] = {"params": {"decay": 0.5, "update_every": 1}}
    first = Trainer(cfg, tiny_model(), FixedBatch(), noise_scheduler=FlowMatchEulerDiscreteScheduler())
    first.train_loop()
    ckpt = tmp_path / "checkpoints" / "checkpoint-4"
    ema_saved = {k: v.clone() for k, v in first.ema.model.state_dict().items()}
    raw_saved = first.model.state_dict()
    assert any(not torch.equal(ema_saved[k].float(), raw_saved[k].float()) for k in ema_saved)  # the average lags the weights
    cfg2 = tiny_config(tmp_path, steps=5)
    cfg2["ema"] = cfg["ema"]
    cfg2["experiment"].update(resume_from_checkpoint=str(ckpt), resume_iter=4)
    resumed = Trainer(cfg2, NOVATransformer3DModel.from_pretrained(str(ckpt / "transformer")), FixedBatch(),
                      noise_scheduler=FlowMatchEulerDiscreteScheduler())
    for k, v in resumed.ema.model.state_dict().items():
        assert torch.equal(v.float(), ema_saved[k].float()), k
    # the reference's on-disk format: a torch.save'd state_dict named diffusion_pytorch_model.bin
    legacy = tmp_path / "legacy"
    legacy.mkdir()
    (legacy / "config.json").write_text((ckpt / "transformer" / "config.json").read_text())
    torch.save(dict(raw_saved), legacy / "diffusion_pytorch_model.bin")
    again = NOVATransformer3DModel.from_pretrained(str(legacy))
    for k, v in again.state_dict().items():
        assert torch.equal(v, raw_saved[k]), k


def test_train_script_runs_and_resumes(tmp_path):
    import yaml

    import train as train_script

    tiny_model()  # registers the toy architectures
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "train_pointcloud_tiny.yaml")))
    cfg["experiment"]["output_dir"] = str(tmp_path)
    cfg["model"]["params"].update(image_size=[64, 64], image_base_size=[4, 4], video_base_size=[1, 2, 2], text_token_dim=32,
                                  arch=["tr_vit_d1w64", "tr_vit_d2w64", "tr_mlp_d1w64"])
    cfg["training"].update(max_train_steps=2, device="cpu")
    path = tmp_path / "cfg.yaml"
    yaml.safe_dump(cfg, open(path, "w"))
    history, ckpt = train_script.main(["--config", str(path)])
    assert len(history) == 2 and os.path.isdir(ckpt) and ckpt.endswith(os.path.join("checkpoint-2", "transformer"))
    history2, ckpt2 = train_script.main(["--config", str(path), "experiment.resume_from_checkpoint=latest", "training.max_train_steps=3"])
    assert [h["step"] for h in history2] == [2] and ckpt2.endswith(os.path.join("checkpoint-3", "transformer"))


def _dp_worker(rank, world, port, out_dir, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from diffnext.engine import engine_utils
    from diffnext.engine.datasets import SyntheticPointClouds
    from diffnext.engine.train_engine import Trainer
    from diffnext.schedulers import FlowMatchEulerDiscreteScheduler

    model = tiny_model(seed=100 + rank)  # DIFFERENT initial weights per rank: the reducer must broadcast rank 0's
    engine_utils.manual_seed(7 + rank)
    loader = SyntheticPointClouds(2, (4, 4), 32, seed=5, shard_id=rank, num_shards=world)
    trainer = Trainer(tiny_config(out_dir, steps=3, accum=2), model, loader, noise_scheduler=FlowMatchEulerDiscreteScheduler())
    assert len(trainer.reducer.buckets) > 1
    # one manual step: local gradients, then the exchange, checked against the mean of both ranks' local gradients
    metrics = {"loss": 0.0}
    inputs = loader.next()[0]
    trainer.model(inputs)["loss"].backward()
    local = torch.cat([p.grad.reshape(-1) for p in trainer.reducer.params])
    both = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    trainer.reducer.sync_gradients()
    synced = torch.cat([p.grad.reshape(-1) for p in trainer.reducer.params])
    mean_ok = bool((synced - sum(both) / world).abs().max() <= 1e-7 * (1 + synced.abs().max()))
    differ = bool((both[0] - both[1]).abs().max() > 0)  # the ranks really saw different data / draws
    trainer.optimizer.zero_grad(set_to_none=True)
    history = trainer.train_loop()
    flat = torch.cat([p.detach().reshape(-1) for p in trainer.model.parameters()])
    got = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(got, flat)
    if rank == 0:
        ret.put(dict(mean_ok=mean_ok, differ=differ, same=bool(torch.equal(got[0], got[1])), steps=len(history),
                     loss=[h["metrics"]["loss"] for h in history], ckpt=os.path.isdir(os.path.join(out_dir, "checkpoints", "checkpoint-2"))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_training_keeps_ranks_identical(tmp_path):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, str(tmp_path), ret)) for r in range(2)]
    [p.start() for p in procs]
    got = ret.get(timeout=300)
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert got["mean_ok"] and got["differ"] and got["same"] and got["steps"] == 3 and got["ckpt"]
    assert all(math.isfinite(v) for v in got["loss"])


def test_npy_dataset_reads_exported_point_clouds(tmp_path):
    """The .npy files `save_point_clouds` writes (README.md:108-113 layout) are a training set: subsampled to the canvas."""
    import numpy as np

    from diffnext.engine.datasets import NpyPointClouds
    from nova_pointcloud_amd.metrics import save_point_clouds

    pts = torch.randn(3, 40, 3)
    save_point_clouds(pts, "shape", str(tmp_path))
    np.save(tmp_path / "shape_1.prompt.npy", np.ones((12, 32), "float32"))
    ds = NpyPointClouds(str(tmp_path), batch_size=3, latent_hw=(4, 4), token_dim=32, max_prompt_len=8)
    batch = ds.next()[0]
    assert batch["x"].shape == (3, 3, 4, 4) and [p.shape for p in batch["prompt"]] == [(1, 32), (8, 32), (1, 32)]
    rows = batch["x"][0].reshape(3, -1).t()
    assert all(any(torch.equal(r, q) for q in pts[0]) for r in rows)  # every token is one of the sample's points
    shard = NpyPointClouds(str(tmp_path), batch_size=1, latent_hw=(4, 4), token_dim=32, shard_id=1, num_shards=2)
    assert len(shard.files) == 1 and shard.files[0].endswith("shape_1.npy")
    with pytest.raises(ValueError, match="Unsupported dataset"):
        NpyPointClouds(str(tmp_path / "none"), 1, (4, 4), 32)
