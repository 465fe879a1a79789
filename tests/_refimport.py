"""Import the reference's real `models/graph_gen.py` (read-only, only present
in the build container) under empty `tensorflow` / `open3d` stubs.  Its
top-level imports (graph_gen.py:8-9) are unused by the functions on the hot
path.  Returns None when /root/reference is absent (GPU box)."""
import os
import sys
import types

REF_ROOT = "/root/reference"


def reference_graph_gen():
    if not os.path.isdir(os.path.join(REF_ROOT, "models")):
        return None
    for name in ("tensorflow", "open3d"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    try:
        from models import graph_gen  # noqa: the reference's module
    finally:
        sys.path.remove(REF_ROOT)
    return graph_gen
