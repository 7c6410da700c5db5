"""Iterative training of the NOVA generator (reference diffnext/engine/train_engine.py:32-175, scripts/train.py:29-101,
diffnext/pipelines/nova/pipeline_train_t2i.py:27-93).

Same schedule of work per iteration as the reference's `Trainer`: scheduled learning rate -> `accum_steps` micro-steps of
forward (`Transformer3DModel.train_video`) + backward -> optimizer step -> zero_grad -> scheduler step; logging, EMA and
`checkpoint-<step>/<name>` snapshots on their periods. What differs is the plumbing the reference takes from packages that
are absent here (accelerate, omegaconf, wandb, codewithgpu): the configuration is a plain nested dict (scripts/train.py
reads YAML), data parallelism is one process per GPU with bucketed gradient all-reduce over RCCL
(`data_parallel.GradientReducer`), the data source is any object with `.next() -> [inputs]`.

Compute: with autograd enabled the modules run their PyTorch definitions on the GPU (the hand-written HIP kernels are
forward-only), so a training step is numerically the reference's own.
"""
import collections
import logging
import os
import time

import torch

from . import engine_utils
from .data_parallel import GradientReducer
from .lr_scheduler import ConstantLR, CosineLR, MultiStepLR

LR_SCHEDULERS = {"ConstantLR": ConstantLR, "CosineLR": CosineLR, "MultiStepLR": MultiStepLR}


def configure_model(model, config, noise_scheduler=None):
    """NOVATrainT2IPipeline.configure_model (pipeline_train_t2i.py:57-72): checkpointing levels, loss_repeat, the modules
    that stay frozen during text-to-sample training, the 'trainable model' guard of the preprocess hook."""
    mcfg = config.get("model", {})
    level = mcfg.get("gradient_checkpointing", 0)
    model.loss_repeat = mcfg.get("loss_repeat", 4)
    for blk in model.video_encoder.blocks:
        blk.mlp_checkpointing = level
    for blk in model.image_encoder.blocks:
        blk.mlp_checkpointing = level > 1
    for blk in model.image_decoder.blocks:
        blk.mlp_checkpointing = level > 2
    engine_utils.freeze_module(model.text_embed.norm)
    engine_utils.freeze_module(model.video_pos_embed)
    engine_utils.freeze_module(model.video_encoder.patch_embed)
    if model.motion_embed is not None:
        engine_utils.freeze_module(model.motion_embed)
    if noise_scheduler is not None:
        model.noise_scheduler = noise_scheduler

    def preprocess(inputs):
        if not model.training:
            raise RuntimeError("Excepted a trainable model.")
        return inputs

    model.pipeline_preprocess = preprocess
    model.train()
    model.text_embed.norm.eval(), model.video_pos_embed.eval(), model.video_encoder.patch_embed.eval()
    return model


class SmoothedValue(object):
    def __init__(self):
        self.total, self.count = 0.0, 0

    def update(self, v):
        self.total, self.count = self.total + float(v), self.count + 1

    def average(self):
        return self.total / max(self.count, 1)


class Trainer(object):
    """Schedule the iterative model training.

    config keys (the reference's YAML sections): training.{max_train_steps, gradient_accumulation_steps, seed, max_grad_norm},
    optimizer.{target: AdamW|SGD|..., params}, lr_scheduler.{target, params}, experiment.{output_dir, log_every, save_every,
    resume_iter}, model.{name, loss_repeat, gradient_checkpointing}, ema.params (optional), parallel.bucket_mb.
    """

    def __init__(self, config, model, train_dataloader, logger=None, noise_scheduler=None, process_group=None):
        self.config, self.train_dataloader = config, train_dataloader
        self.logger = logger or logging.getLogger("diffnext.train")
        self.model = configure_model(model, config, noise_scheduler)
        self.ema = engine_utils.ModelEMA(self.model, **config["ema"].get("params", {})) if "ema" in config else None
        exp = config.get("experiment", {})
        if self.ema is not None and exp.get("resume_iter", 0) > 0 and exp.get("resume_from_checkpoint"):
            # a resumed run continues the averaged weights it had written beside the checkpoint (the reference reloads
            # ema_checkpoints/<ckpt>/<model.name> when resume_iter > 0: engine/train_engine.py:52-56), not a fresh copy of the raw ones
            ema_dir = os.path.join(exp["resume_from_checkpoint"].replace("checkpoints", "ema_checkpoints"),
                                   config.get("model", {}).get("name", "transformer"))
            if os.path.isdir(ema_dir):
                saved = type(self.ema.model).from_pretrained(ema_dir).state_dict()
                with torch.no_grad():
                    for k, v in self.ema.model.state_dict().items():
                        v.copy_(saved[k].to(v.dtype))
                self.logger.info("Resumed EMA weights from: %s", ema_dir)
            else:
                self.logger.warning("EMA configured but %s does not exist: averaging restarts from the resumed weights", ema_dir)
        groups = engine_utils.get_param_groups(self.model)
        opt = config.get("optimizer", {"target": "AdamW", "params": {"lr": 1e-4}})
        self.optimizer = getattr(torch.optim, opt["target"].rsplit(".", 1)[-1])(groups, **opt.get("params", {}))
        sch = config.get("lr_scheduler", {"target": "ConstantLR", "params": {"lr_max": opt.get("params", {}).get("lr", 1e-4)}})
        self.scheduler = LR_SCHEDULERS[sch["target"].rsplit(".", 1)[-1]](**sch.get("params", {}))
        self.reducer = GradientReducer(list(self.model.parameters()), config.get("parallel", {}).get("bucket_mb", 256.0), process_group)
        self.metrics = collections.OrderedDict()
        self.is_main_process = (not torch.distributed.is_initialized()) or torch.distributed.get_rank(process_group) == 0

    @property
    def global_step(self) -> int:
        return self.scheduler._step_count

    def save(self):
        exp = self.config.get("experiment", {})
        path = os.path.join(exp.get("output_dir", "."), "checkpoints", f"checkpoint-{self.global_step}",
                            self.config.get("model", {}).get("name", "transformer"))
        if self.is_main_process and not os.path.exists(path):
            self.model.save_pretrained(path)
            self.logger.info("Wrote snapshot to: %s", path)
            if self.ema is not None:
                ema_path = path.replace("checkpoints", "ema_checkpoints")
                self.ema.model.save_pretrained(ema_path)
        return path

    def run_model(self, metrics, accum_steps=1):
        """`accum_steps` micro-steps: every output named *loss* / *metric* is logged (mean over ranks); the losses that
        require grad are summed and back-propagated; gradients are averaged over the ranks after the last micro-step."""
        for _ in range(accum_steps):
            inputs = self.train_dataloader.next()[0]
            outputs, losses = self.model(inputs), []
            for k, v in outputs.items():
                if "loss" not in k and "metric" not in k:
                    continue
                if torch.is_tensor(v) and v.requires_grad:
                    losses.append(v)
                metrics[k] += self.reducer.gather_mean(v.detach()) / accum_steps
            (sum(losses[1:], losses[0]) / accum_steps).backward()
        self.reducer.sync_gradients()

    def run_step(self, accum_steps=1) -> dict:
        stats = {"step": self.global_step}
        metrics = collections.defaultdict(float)
        tic = time.time()
        stats["lr"] = self.scheduler.get_lr()
        for group in self.optimizer.param_groups:
            group["lr"] = stats["lr"] * group.get("lr_scale", 1.0)
        self.run_model(metrics, accum_steps)
        clip = self.config.get("training", {}).get("max_grad_norm", None)
        if clip:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), clip)
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        self.scheduler.step()
        stats["time"] = time.time() - tic
        stats["metrics"] = collections.OrderedDict(sorted(metrics.items()))
        return stats

    def log_metrics(self, stats):
        self.logger.info("Iteration %d, lr = %.8f, time = %.2fs", stats["step"], stats["lr"], stats["time"])
        for k, v in self.metrics.items():
            self.logger.info("    Train net output(%s): %.4f (%.4f)", k, stats["metrics"][k], v.average())
        self.metrics.clear()

    def train_loop(self):
        tr, exp = self.config.get("training", {}), self.config.get("experiment", {})
        max_steps, accum = tr.get("max_train_steps", 1), tr.get("gradient_accumulation_steps", 1)
        log_every, save_every = exp.get("log_every", 10), exp.get("save_every", 10 ** 9)
        self.scheduler._step_count = exp.get("resume_iter", 0)
        history = []
        while self.global_step < max_steps:
            stats = self.run_step(accum)
            history.append(stats)
            for k, v in stats["metrics"].items():
                self.metrics.setdefault(k, SmoothedValue()).update(v)
            if stats["step"] % log_every == 0:
                self.log_metrics(stats)
            if self.ema and self.global_step % self.ema.update_every == 0:
                self.ema.update(self.model)
            if self.global_step % save_every == 0:
                self.save()
        return history
