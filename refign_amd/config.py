"""YAML `class_path` / `init_args` instantiation so that the reference's configs/*.yaml run unmodified.

The reference drives everything through LightningCLI + jsonargparse (tools/run.py:4-9, helpers/cli.py:10-21): every
object is a `{class_path: pkg.Class, init_args: {...}}` dict, nested recursively; `optimizer` and `lr_scheduler` are
linked into `model.init_args.{optimizer_init, lr_scheduler_init}` as dicts and instantiated later by
`instantiate_class` (segmentation_model.py:385-387).  Neither package is installed here, so this file provides the same
two operations.  Class paths of the reference's own packages are routed to their MI355X implementations.
"""
import importlib

import yaml

# reference class path -> implementation in this repo
ALIASES = {
    "models.DomainAdaptationSegmentationModel": "refign_amd.uda.DomainAdaptationSegmentationModel",
    "models.AlignmentModel": "refign_amd.alignment_model.AlignmentModel",
    "models.backbones.MixVisionTransformer": "refign_amd.seg.MixVisionTransformer",
    "models.backbones.VGG": "refign_amd.align.VGG",
    "models.heads.DAFormerHead": "refign_amd.seg.DAFormerHead",
    "models.heads.SegFormerHead": "refign_amd.seg.SegFormerHead",
    "models.heads.UAWarpCHead": "refign_amd.align.UAWarpCHead",
    "models.losses.PixelWeightedCrossEntropyLoss": "refign_amd.seg.PixelWeightedCrossEntropyLoss",
    "models.losses.HuberLoss": "refign_amd.losses.HuberLoss",
    "models.losses.MultiScaleFlowLoss": "refign_amd.losses.MultiScaleFlowLoss",
    "models.losses.WBipathLoss": "refign_amd.losses.WBipathLoss",
    "helpers.metrics.IoU": "refign_amd.metrics.IoU",
    "helpers.metrics.SparseEPE": "refign_amd.metrics.SparseEPE",
    "helpers.lr_scheduler.LinearWarmupPolynomialLR": "refign_amd.trainer.LinearWarmupPolynomialLR",
    "helpers.callbacks.ValEveryNSteps": "refign_amd.trainer.ValEveryNSteps",
}
# subsystems that are out of scope here (host I/O, logging, evaluation): accepted in a config, not instantiated
IGNORED_PREFIXES = ("pytorch_lightning.", "data_modules.", "helpers.metrics.")
# class paths accepted in a config but kept as specs (none at present: the matcher-training losses of row N1 are built)
DEFERRED = ()


# reference packages whose classes are routed through ALIASES; anything of theirs that has no alias is a component
# outside the align-and-refine hot path (ResNet / DeepLabV2, data modules, metrics, ...)
_REFERENCE_PACKAGES = ("models.", "helpers.", "data_modules.")


class OutOfScopeError(NotImplementedError):
    pass


def resolve(class_path):
    path = ALIASES.get(class_path)
    if path is None:
        if class_path.startswith(_REFERENCE_PACKAGES):
            raise OutOfScopeError(f"{class_path} is out of scope of refign_amd (the align-and-refine hot path: "
                                  f"{', '.join(sorted(ALIASES))})")
        path = class_path
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def is_spec(x):
    return isinstance(x, dict) and "class_path" in x


def build(spec):
    """Recursively instantiate a {class_path, init_args} tree (nested specs inside init_args are built first)."""
    if is_spec(spec):
        if spec["class_path"].startswith(IGNORED_PREFIXES) or spec["class_path"] in DEFERRED:
            return spec
        kwargs = {k: build(v) for k, v in spec.get("init_args", {}).items()}
        return resolve(spec["class_path"])(**kwargs)
    if isinstance(spec, list):
        return [build(v) for v in spec]
    return spec


def instantiate_class(args, init):
    """pytorch_lightning.utilities.cli.instantiate_class: `init` = {class_path, init_args}, `args` = positional."""
    args = args if isinstance(args, tuple) else (args,)
    return resolve(init["class_path"])(*args, **init.get("init_args", {}))


def load_config(path):
    with open(path) as f:
        return yaml.safe_load(f)


def build_model(cfg, overrides=None):
    """Build `cfg['model']` with the optimizer / lr_scheduler sections linked in as dicts (helpers/cli.py:17-21).
    `overrides` (dict) is merged into model.init_args first, e.g. {'backbone.init_args.pretrained': None}."""
    model = dict(cfg["model"])
    init = dict(model.get("init_args", {}))
    init["optimizer_init"] = cfg.get("optimizer", init.get("optimizer_init"))
    init["lr_scheduler_init"] = cfg.get("lr_scheduler", init.get("lr_scheduler_init"))
    for dotted, val in (overrides or {}).items():
        node = init
        keys = dotted.split(".")
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = val
    # metrics are evaluation-only (helpers.metrics, torchmetrics): kept as config, not built
    metrics = init.pop("metrics", {})
    kwargs = {k: (v if k in ("optimizer_init", "lr_scheduler_init") else build(v)) for k, v in init.items()}
    return resolve(model["class_path"])(metrics=metrics, **kwargs)
