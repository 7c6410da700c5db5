"""`diffnext` — drop-in for the reference package of the same name, hot path on MI355X.

Same import paths, class names, constructor signatures and state_dict keys as
zailaiyiwan123/NOVA_pointcloud's `diffnext` for the generation path
(`diffnext.pipelines.NOVAPipeline`, `diffnext.models.*`, `diffnext.schedulers.*`). With a model
on an MI355X, `NOVAPipeline.__call__` runs the autoregressive loop on the hand-written gfx950
kernels of libnova_hip.so (nova_pointcloud_amd/engine.py); there is no eager fallback there.
On CPU tensors, or with autograd enabled (training), the modules run their PyTorch definition.

Put `<repo>/nova_pointcloud_amd` in front of PYTHONPATH to shadow the reference's package.
"""
__version__ = "0.1.0"
