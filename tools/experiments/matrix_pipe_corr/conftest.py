"""fixtures of the product's test suite (dev, oracle, the gpu marker) for the tool tests of this directory"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests"))
_here = sys.modules.pop("conftest", None)
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("_product_conftest", os.path.join(sys.path[0], "conftest.py"))
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
if _here is not None:
    sys.modules["conftest"] = _here
