#!/usr/bin/env python
"""Every frame of the bench pools through the three edge arithmetics: max
|dlogit| / |dbox| of 'bf16x3' and 'f16x2' against the fp32 path, the f16x2
range flag, and bit-identity of a repeated run (the parity tests pin seed 0 of
each preset against the float64 oracle; this is the sweep over the rest).

    python tools/arith_soak.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import configs, weights  # noqa: E402
from pointgnn_amd.engine import InferenceEngine  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402

dev = torch.device("cuda", 0)
worst = 0.0
for config, preset, seeds in (("car_auto_T3", "car_600k", range(8)),
                              ("car_auto_T3", "car", range(4)),
                              ("ped_cyl_auto_T3", "ped_dense", range(4))):
    cfg = configs.get_config(config)
    params = weights.init_params(cfg, seed=0, bias_scale=0.05)
    eng = InferenceEngine(cfg, params, device=dev)
    for seed in seeds:
        xyz, inten = synthetic_cloud(seed=seed, preset=preset)
        x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(inten).to(dev)
        out = {}
        for arith in ("f32", "bf16x3", "f16x2"):
            eng.model.edge_arith = arith
            lg, bx = [t.clone() for t in eng.run_frame(x, f)]
            lg2, bx2 = eng.run_frame(x, f)
            assert torch.equal(lg, lg2) and torch.equal(bx, bx2), \
                "%s is not deterministic" % arith
            out[arith] = (lg, bx)
        ok = eng.model.edge_range_ok()
        eng.model.edge_arith = "f32"
        k = int(out["f32"][0].shape[0])
        line = "%s/%s seed %d K %5d |logit|max %.3g:" % (
            config, preset, seed, k, float(out["f32"][0].abs().max()))
        for arith in ("bf16x3", "f16x2"):
            dl = float((out[arith][0] - out["f32"][0]).abs().max())
            db = float((out[arith][1] - out["f32"][1]).abs().max())
            worst = max(worst, dl, db)
            line += "  %s dlogit %.3g dbox %.3g" % (arith, dl, db)
        print(line + ("" if ok else "  RANGE FLAG"))
        assert ok
print("worst difference to the fp32 path: %.3g" % worst)
assert worst < 1e-5
