#!/usr/bin/env python
"""ped_cyl frames through the whole-frame-per-stream schedule, 6 x 32 frames:
the caching allocator's reserve stays flat (12.1 GB with the split pooling
stage's hidden-row workspaces of three frames in flight).

    python tools/micro/ped_soak.py
"""
import torch, sys
sys.path.insert(0, ".")
import pointgnn_amd
from pointgnn_amd import configs, weights
from pointgnn_amd.engine import InferenceEngine
from pointgnn_amd.synthetic import synthetic_cloud
dev = torch.device("cuda", 0)
cfg = configs.get_config("ped_cyl_auto_T3")
eng = InferenceEngine(cfg, weights.init_params(cfg, seed=0, bias_scale=0.05), device=dev)
fr = []
for s in range(4):
    x, f = synthetic_cloud(seed=s, preset="ped_dense")
    fr.append((torch.from_numpy(x).to(dev), torch.from_numpy(f).to(dev)))
for it in range(6):
    out = eng.run_frames_on_streams(fr * 8, 3)
    torch.cuda.synchronize()
    print(it, "reserved GB %.2f allocated GB %.2f" % (torch.cuda.memory_reserved() / 1e9, torch.cuda.memory_allocated() / 1e9))
