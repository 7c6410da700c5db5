"""Train a NOVA point-set generator on MI355X (the entry point of reference scripts/train.py:87-101).

    python scripts/train.py --config configs/train_pointcloud.yaml [key=value ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train.py --config ...

One process per GPU; gradients are averaged over RCCL (xGMI) in large flat buckets. Seeds are `training.seed + rank`, the
dataset is sharded by rank, rank 0 writes `checkpoint-<step>/<model.name>` and `config.yaml` under experiment.output_dir
and resumes from the latest checkpoint when experiment.resume_from_checkpoint = latest.
"""
import argparse
import logging
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nova_pointcloud_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def prepare_checkpoints(config):
    """Resolve experiment.resume_from_checkpoint ('latest' or a checkpoint-<n> path) to (path, resume_iter)."""
    exp = config.setdefault("experiment", {})
    exp.setdefault("resume_from_checkpoint", "")
    ckpt_dir = os.path.abspath(os.path.join(exp.get("output_dir", "."), "checkpoints"))
    os.makedirs(ckpt_dir, exist_ok=True)
    resume_iter = 0
    if exp["resume_from_checkpoint"] == "latest":
        found = sorted((int(n.split("-")[-1]), n) for n in os.listdir(ckpt_dir) if n.startswith("checkpoint-"))
        exp["resume_from_checkpoint"] = os.path.join(ckpt_dir, found[-1][1]) if found else ""
        resume_iter = found[-1][0] if found else 0
    elif exp["resume_from_checkpoint"]:
        resume_iter = int(os.path.basename(os.path.normpath(exp["resume_from_checkpoint"])).split("-")[-1])
    exp["resume_iter"] = resume_iter


def set_by_path(config, dotted, value):
    node = config
    keys = dotted.split(".")
    for k in keys[:-1]:
        node = node.setdefault(k, {})
    node[keys[-1]] = yaml.safe_load(value)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("overrides", nargs="*", help="dotted.key=value")
    args = ap.parse_args(argv)
    with open(args.config) as f:
        config = yaml.safe_load(f)
    for item in args.overrides:
        k, v = item.split("=", 1)
        set_by_path(config, k, v)

    import torch.distributed as dist

    from diffnext.engine import engine_utils
    from diffnext.engine.datasets import NpyPointClouds, SyntheticPointClouds
    from diffnext.engine.train_engine import Trainer
    from diffnext.models.transformers.transformer_nova import NOVATransformer3DModel
    from diffnext.schedulers import FlowMatchEulerDiscreteScheduler

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    use_gpu = torch.cuda.is_available() and config.get("training", {}).get("device", "cuda") != "cpu"
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(local)
    # one process per GPU under torchrun; a single-rank launch under torchrun also goes through the process group, so the
    # RCCL path (bucketed all_reduce of the gradients, barrier) is exercised on a one-GPU box as well
    distributed = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl" if use_gpu else "gloo", rank=rank, world_size=world, **({"device_id": device} if use_gpu else {}))
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format="%(asctime)s %(message)s")
    logger = logging.getLogger("diffnext.train")

    seed = config.setdefault("training", {}).get("seed", 1337) + rank
    engine_utils.manual_seed(seed, (local, seed) if use_gpu else None)
    prepare_checkpoints(config)
    mcfg = config["model"]
    exp = config["experiment"]
    if exp["resume_from_checkpoint"]:
        model = NOVATransformer3DModel.from_pretrained(os.path.join(exp["resume_from_checkpoint"], mcfg.get("name", "transformer")))
    else:
        torch.manual_seed(config["training"].get("seed", 1337))  # identical initial weights on every rank
        model = NOVATransformer3DModel(**mcfg["params"])
        engine_utils.manual_seed(seed, (local, seed) if use_gpu else None)
    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "no": torch.float32}[config["training"].get("mixed_precision", "no")]
    model = model.to(device=device, dtype=dtype)
    data = config["train_dataloader"]
    latent = [s // mcfg["params"]["image_stride"] for s in mcfg["params"]["image_size"]]
    dargs = dict(batch_size=data.get("batch_size", 4), latent_hw=latent, token_dim=mcfg["params"]["text_token_dim"],
                 seed=config["training"].get("seed", 1337), device=device, dtype=dtype, shard_id=rank, num_shards=world,
                 max_prompt_len=mcfg["params"]["text_token_len"])
    loader = NpyPointClouds(data["dataset"], **dargs) if data.get("dataset") else SyntheticPointClouds(**dargs)
    trainer = Trainer(config, model, loader, logger, noise_scheduler=FlowMatchEulerDiscreteScheduler())
    logger.info("#Params: %.2fM", engine_utils.count_params(trainer.model))
    if rank == 0:
        os.makedirs(exp.get("output_dir", "."), exist_ok=True)
        with open(os.path.join(exp.get("output_dir", "."), "config.yaml"), "w") as f:
            yaml.safe_dump(config, f)
    history = trainer.train_loop()
    if trainer.ema:
        trainer.ema.update(trainer.model)
    path = trainer.save()
    if rank == 0 and os.environ.get("NOVA_TRAIN_LOG_JSON"):  # machine-readable summary for the launch tests
        import json

        with open(os.environ["NOVA_TRAIN_LOG_JSON"], "w") as f:
            json.dump({"world": world, "backend": dist.get_backend() if distributed else None, "dtype": str(dtype),
                       "loss": [h["metrics"]["loss"] for h in history], "hip_attention_calls": _hip_attention_calls()}, f)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return history, path



def _hip_attention_calls():
    """Forward calls of the HIP training attention in this process (0 on the PyTorch path / without the library)."""
    mod = sys.modules.get("nova_pointcloud_amd.autograd")
    return mod.stats["attention_calls"] if mod is not None else 0


if __name__ == "__main__":
    main()
