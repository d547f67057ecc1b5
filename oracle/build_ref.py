"""oracle/build_ref.py -- TEST INFRASTRUCTURE.

Compiles the REFERENCE's own CPU correlation sampler from the sources where they lie
(/root/reference/models/correlation_ops/{correlation.cpp,correlation_sampler_cpu.cpp})
into oracle/_ref/ (git-ignored, travels to the GPU box as a prebuilt .so).  Nothing is copied
into the repo; the reference's own build (JIT into its source dir, __init__.py:6-30) is not run.

The result is an ordinary torch C++ extension module exposing `forward` / `backward`
(correlation_sampler_cpu.cpp:34-37).  It is used (a) to pin oracle/corr_oracle.c and to generate
tests/golden/corr_*.npz, (b) optionally as bench.py's cpu_baseline kind="reference".
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = "/root/reference/models/correlation_ops"
OUT_DIR = os.path.join(HERE, "_ref")
NAME = "refign_reference_correlation"


def build(verbose=False):
    if not os.path.isdir(REF_DIR):
        raise FileNotFoundError(f"{REF_DIR} not present (only available in the authoring container)")
    from torch.utils import cpp_extension
    os.makedirs(OUT_DIR, exist_ok=True)
    return cpp_extension.load(
        NAME,
        sources=[os.path.join(REF_DIR, "correlation.cpp"),
                 os.path.join(REF_DIR, "correlation_sampler_cpu.cpp")],
        build_directory=OUT_DIR,
        extra_cflags=["-O2", "-fopenmp"],
        extra_ldflags=["-lgomp"],
        with_cuda=False,
        verbose=verbose)


def load_prebuilt():
    """Import oracle/_ref/<NAME>.so without compiling (GPU box: /root/reference is absent)."""
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    so = os.path.join(OUT_DIR, NAME + ".so")
    if not os.path.exists(so):
        return None
    spec = importlib.util.spec_from_file_location(NAME, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    m = build(verbose="-v" in sys.argv)
    print("built", m)
