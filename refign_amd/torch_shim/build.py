"""refign_amd/torch_shim/build.py -- builds the pybind11 / torch-extension module `correlation` (correlation_shim.cpp) IN-TREE
(refign_amd/torch_shim/correlation.so) against librefign_hip.so; `load()` imports it.  The reference loads its own module the
same way (models/correlation_ops/__init__.py:6-30: cpp_extension.load into the package directory); a maintainer replaces that
call with `from refign_amd.torch_shim.build import load; correlation = load()` (option C)."""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(os.path.dirname(HERE), "lib")
SO = os.path.join(HERE, "correlation.so")


def build(verbose=False):
    src = os.path.join(HERE, "correlation_shim.cpp")
    lib = os.path.join(LIBDIR, "librefign_hip.so")
    if os.path.exists(SO) and os.path.getmtime(SO) >= max(os.path.getmtime(src), os.path.getmtime(lib) if os.path.exists(lib) else 0):
        return SO
    from torch.utils import cpp_extension
    bdir = os.path.join(HERE, "_build")
    os.makedirs(bdir, exist_ok=True)
    import fcntl
    with open(os.path.join(bdir, ".lock"), "w") as lock:          # one builder at a time (ranks of one node share the tree)
        fcntl.flock(lock, fcntl.LOCK_EX)
        if os.path.exists(SO) and os.path.getmtime(SO) >= max(os.path.getmtime(src), os.path.getmtime(lib) if os.path.exists(lib) else 0):
            return SO
        _compile(cpp_extension, src, bdir, verbose)
        os.replace(os.path.join(bdir, "correlation.so"), SO)
    return SO


def _compile(cpp_extension, src, bdir, verbose):
    cpp_extension.load(
        name="correlation", sources=[src], build_directory=bdir, verbose=verbose, with_cuda=False, is_python_module=False,
        extra_cflags=["-O2", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"],
        extra_include_paths=["/opt/rocm/include"],
        extra_ldflags=[f"-L{LIBDIR}", "-lrefign_hip", "-Wl,-rpath,\\$$ORIGIN/../lib",
                       "-Wl,-rpath,\\$$ORIGIN/../../lib", "-L/opt/rocm/lib", "-lamdhip64",
                       "-lc10_hip", "-ltorch_hip"])


def load():
    """-> the extension module (attributes `forward`, `backward`); builds it first if needed."""
    import torch  # noqa: F401  (libtorch must be loaded before the extension)
    path = build()
    spec = importlib.util.spec_from_file_location("correlation", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
