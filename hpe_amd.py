"""Import alias: ``import hpe_amd`` == the package directory ``human-pose-estimation_amd`` (whose name
is not a Python identifier)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("human-pose-estimation_amd")
