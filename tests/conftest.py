import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# float64 oracle outputs of the BASELINE-size frames, shared by the test
# modules that compare different device paths with them on the SAME
# device-built graph (the oracle run of `ped_dense` alone takes ~30 s of host
# time): key -> (edge digests, logits, boxes, features)
_FULLSIZE_ORACLE = {}


def fullsize_oracle(key, params, cfg, inten, c_np, k_np, e_np):
    """gn.predict(..., dtype=float64, return_features=True), computed once per
    (key, graph): the cache entry is only reused for byte-identical edge
    lists."""
    import hashlib
    import numpy as np
    from oracle import gnn_oracle as gn
    sig = tuple(hashlib.sha256(np.ascontiguousarray(e).tobytes()).hexdigest()
                for e in list(e_np) + list(k_np))
    hit = _FULLSIZE_ORACLE.get(key)
    if hit is not None and hit[0] == sig:
        return hit[1:]
    lg, bx, feats = gn.predict(params, cfg, inten, c_np, k_np, e_np,
                               dtype=np.float64, return_features=True)
    _FULLSIZE_ORACLE[key] = (sig, lg, bx, feats)
    return lg, bx, feats


# The arithmetic of the per-edge product (model.edge_arith): the parity tests
# that hold the fp32-MFMA path to the reference run again on each SECONDARY
# arithmetic (split-bf16, two-part fp16).  Small fixtures have fewer edges than the
# kernel's launcher asks for (it would answer "unsupported" and the fp32 entry
# would run): the `b16_force` tunable lifts that for the duration of the test.
EDGE_ARITHS = ("f32", "bf16x3", "f16x2")


@pytest.fixture(params=EDGE_ARITHS)
def edge_arith(request):
    if request.param == "f32":
        yield "f32"
        return
    from pointgnn_amd import _lib
    _lib.set_tunable("b16_force", 1)
    try:
        yield request.param
    finally:
        _lib.set_tunable("b16_force", 0)
