"""tests/golden/_ref_import.py -- authoring-container-only helper (NOT used by any test at run time).

Imports the *reference* Python modules from /root/reference so that make_golden.py can emit golden
input/output vectors.  The reference cannot be imported as a package here (models/__init__.py pulls
pytorch_lightning, which is not installed), so the package __init__ files are bypassed by
pre-registering empty package modules, and the missing third-party packages get throw-away stubs
(SURVEY.md §8c).  Nothing from the reference is copied; only tensors it computes are saved.
"""
import importlib
import os
import sys
import types

REF = "/root/reference"


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def setup():
    if not os.path.isdir(REF):
        raise FileNotFoundError("reference not present; goldens can only be regenerated in the authoring container")
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import torch.nn as nn

    # --- third-party stubs ------------------------------------------------------------------
    class _Registry:
        def __call__(self, cls):
            return cls

    def instantiate_class(args, init):
        raise NotImplementedError

    pl = _stub("pytorch_lightning", LightningModule=nn.Module, Trainer=object, Callback=object)
    _stub("pytorch_lightning.utilities")
    _stub("pytorch_lightning.utilities.cli", MODEL_REGISTRY=_Registry(), instantiate_class=instantiate_class)
    pl.utilities = sys.modules["pytorch_lightning.utilities"]

    class _Metric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def add_state(self, *a, **k):
            pass

    class _MetricCollection(nn.ModuleDict):
        def __init__(self, metrics=None, *a, **k):
            super().__init__(metrics or {})

    _stub("torchmetrics", Metric=_Metric, MetricCollection=_MetricCollection, JaccardIndex=_Metric)
    _stub("torchmetrics.functional")
    _stub("torchmetrics.functional.classification")
    _stub("torchmetrics.functional.classification.confusion_matrix", _confusion_matrix_update=None)
    _stub("torchmetrics.utilities")
    _stub("torchmetrics.utilities.data", dim_zero_cat=None)
    _stub("kornia")
    _stub("kornia.augmentation", ColorJitter=None)
    _stub("kornia.filters", GaussianBlur2d=None)
    sys.modules["kornia"].augmentation = sys.modules["kornia.augmentation"]
    sys.modules["kornia"].filters = sys.modules["kornia.filters"]

    # --- package bypass ---------------------------------------------------------------------
    _pkg("models", os.path.join(REF, "models"))
    _pkg("models.heads", os.path.join(REF, "models", "heads"))
    _pkg("models.backbones", os.path.join(REF, "models", "backbones"))
    cops = _pkg("models.correlation_ops", os.path.join(REF, "models", "correlation_ops"))

    # the compiled reference correlation (oracle/_ref) stands in for the JIT build the reference's
    # own correlation_ops/__init__.py would do into its (read-only) source dir
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
    import build_ref
    cops.correlation = build_ref.load_prebuilt() or build_ref.build()
    return cops.correlation


def ref_module(name):
    return importlib.import_module(name)
