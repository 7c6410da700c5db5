"""Pipeline directories on disk (reference diffnext/pipelines/builder.py:31-125).

A pipeline is a directory with `model_index.json` and one sub-directory per component (`transformer/`, `scheduler/`,
`vae/`, `text_encoder/`, `tokenizer/`). `get_pipeline_path` assembles a loading directory out of a pretrained one
plus replacement component paths and replacement component configs, by symlinks, exactly as the training scripts of the
reference expect (`configs/*.yaml: pipeline.module_dict / module_config`); `build_pipeline` loads one;
`build_diffusion_scheduler` instantiates the noise / sampling scheduler named in a scheduler directory's config.
No network: every path is local.
"""
import json
import os
import tempfile

import torch


def _plain(obj):
    """JSON default for config containers (omegaconf's DictConfig / ListConfig when that package is installed)."""
    try:
        import omegaconf

        if isinstance(obj, (omegaconf.ListConfig, omegaconf.DictConfig)):
            return omegaconf.OmegaConf.to_container(obj, resolve=True)
    except ImportError:
        pass
    if hasattr(obj, "items"):
        return dict(obj.items())
    if isinstance(obj, (tuple, set)):
        return list(obj)
    raise TypeError(f"{type(obj).__name__} is not JSON serialisable")


def get_pipeline_path(pretrained_path, module_dict: dict = None, module_config: dict = None, target_path: str = None) -> str:
    """The directory to load the pipeline from.

    With neither `module_dict` nor `module_config` this is `pretrained_path` itself. Otherwise a directory (a fresh
    temporary one unless `target_path` is given) is filled with links to every file of every component directory of
    `pretrained_path`; then, for each `name: path` of `module_dict`, the component `name` is replaced by a link to `path`
    (an empty path drops the component from `model_index.json`; the key `model_index` names the index file to start
    from), and for each `name: config` of `module_config` the component's `config.json` is replaced by that config.
    """
    if module_dict is None and module_config is None:
        return pretrained_path
    target_path = target_path or tempfile.mkdtemp()
    for comp in sorted(os.listdir(pretrained_path)):
        src = os.path.join(pretrained_path, comp)
        if not os.path.isdir(src):
            continue
        os.makedirs(os.path.join(target_path, comp), exist_ok=True)
        for name in os.listdir(src):
            os.symlink(os.path.join(src, name), os.path.join(target_path, comp, name))
    replacements = dict(module_dict or {})
    index_file = replacements.pop("model_index", os.path.join(pretrained_path, "model_index.json"))
    with open(index_file) as f:
        index = json.load(f)
    for comp, path in replacements.items():
        if not path:
            index.pop(comp)
            continue
        try:
            os.symlink(path, os.path.join(target_path, comp))
        except FileExistsError:  # the component directory was linked from the pretrained path already
            pass
    for comp, cfg in (module_config or {}).items():
        if not cfg:
            continue
        cfg_file = os.path.join(target_path, comp, "config.json")
        if os.path.lexists(cfg_file):
            os.remove(cfg_file)
        with open(cfg_file, "w") as f:
            json.dump(cfg, f, default=_plain)
    with open(os.path.join(target_path, "model_index.json"), "w") as f:
        json.dump(index, f)
    return target_path


def build_diffusion_scheduler(scheduler_path, sample=False, **kwargs):
    """Scheduler instance from a scheduler directory (its config names the class under `_noise_class_name` /
    `_sample_class_name`) or a copy of a scheduler object's configuration; None for anything else."""
    from ..schedulers.scheduling_cfm import FlowMatchEulerDiscreteScheduler
    from ..schedulers.scheduling_ddpm import DDPMScheduler

    classes = {"FlowMatchEulerDiscreteScheduler": FlowMatchEulerDiscreteScheduler, "DDPMScheduler": DDPMScheduler}
    if isinstance(scheduler_path, str):
        with open(os.path.join(scheduler_path, DDPMScheduler.config_name)) as f:
            name = json.load(f)["_{}_class_name".format("sample" if sample else "noise")]
        return classes[name].from_pretrained(scheduler_path, **kwargs)
    if hasattr(scheduler_path, "config"):
        return classes[type(scheduler_path).__name__].from_config(scheduler_path.config)
    return None


def build_pipeline(pretrained_path, pipe_cls, dtype=torch.float16, **kwargs):
    """`pipe_cls.from_pretrained(pretrained_path, torch_dtype=dtype, trust_remote_code=True, **kwargs)`."""
    kwargs.setdefault("trust_remote_code", True)
    kwargs.setdefault("torch_dtype", dtype)
    return pipe_cls.from_pretrained(pretrained_path, **kwargs)
