import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pointgnn_amd
from pointgnn_amd import configs, weights
from pointgnn_amd.engine import InferenceEngine
from pointgnn_amd.synthetic import synthetic_cloud
from pointgnn_amd import _lib
for kv in sys.argv[3:]:
    k_, v_ = kv.split("=")
    _lib.set_tunable(k_, int(v_))
dev = torch.device("cuda")
name = sys.argv[1] if len(sys.argv) > 1 else "ped_cyl_auto_T3"
preset = sys.argv[2] if len(sys.argv) > 2 else "ped_dense"
cfg = configs.get_config(name)
eng = InferenceEngine(cfg, weights.init_params(cfg, seed=0, bias_scale=0.05), device=dev)
xyz, inten = synthetic_cloud(seed=0, preset=preset)
x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(inten).to(dev)
eng.model.keep_features = True
lg, bx = eng.run_frame(x, f)
graph0 = eng.last_graph
feats0 = [t.clone() for t in eng.model.feature_list]
k = lg.shape[0]
cap = eng.capture_frame(x, f)
cfeats = list(eng.model.feature_list)      # the graph's own buffers
cgraph = eng.last_graph
for rep in range(3):
    out = cap.replay(x, f)
    lg2, bx2 = out.result()
    torch.cuda.synchronize()
    print("rep", rep, "K", lg2.shape[0], "logits equal", torch.equal(lg, lg2), "boxes", torch.equal(bx, bx2))
    for i, (a, b) in enumerate(zip(feats0, cfeats)):
        print("   layer", i, "equal", torch.equal(a, b[:k]), float((a - b[:k]).abs().max()))
    c0, k0, e0 = graph0
    c1, k1, e1 = cgraph
    print("   keypoints equal", torch.equal(c0[1], c1[1][:k]), "kp idx", torch.equal(k0[0], k1[0][:k]),
          "edges0", torch.equal(e0[0], e1[0][:e0[0].shape[0]]), "edges1", torch.equal(e0[1], e1[1][:e0[1].shape[0]]))
