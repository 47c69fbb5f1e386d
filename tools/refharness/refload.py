"""Load the UNMODIFIED reference (/root/reference) in the dev container for golden-vector generation.

Dev-container only: nothing here ships, nothing here is imported by the product, tests or bench.
Two local-only shims:
  (i)  tools/refharness/gymnasium  -- class scaffolding stand-in (gymnasium is absent from the image)
  (ii) a synthetic `core` package entry so that core/__init__.py (which opens the missing
       core/version.txt, core/__init__.py:16-18) is never executed.
"""
import os
import sys
import types

REF = os.environ.get("CSTR_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF} (fixtures can only be regenerated in the dev container)")
    if HERE not in sys.path:
        sys.path.insert(0, HERE)  # stand-in gymnasium
    if REF not in sys.path:
        sys.path.insert(1, REF)  # twoseriescstr.py
    if "core" not in sys.modules:
        core = types.ModuleType("core")
        core.__path__ = [os.path.join(REF, "core")]
        core.__version__ = "2.4.0a-reference"
        sys.modules["core"] = core
    sys.dont_write_bytecode = True
    return sys.modules["core"]
