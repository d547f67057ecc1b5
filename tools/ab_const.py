#!/usr/bin/env python3
"""tools/ab_const.py module.attr=value [...] -- bench.py args...: run bench.py with module constants of refign_amd overridden
(e.g. uda._SRC_BWD_AFTER_TEACHER=False) -- the A/B form of switches that are module attributes, not environment variables."""
import ast
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
split = args.index("--") if "--" in args else len(args)
for item in args[:split]:
    path, val = item.split("=", 1)
    mod, attr = path.rsplit(".", 1)
    setattr(importlib.import_module("refign_amd." + mod), attr, ast.literal_eval(val))
import bench  # noqa: E402
sys.argv = ["bench.py"] + args[split + 1:]
bench.main()
