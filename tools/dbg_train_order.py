"""HIP streams share a few hardware queues: step time of the training bench as
more and more torch streams exist in the process (engine.concurrent_streams
keeps the graph-build stream off the compute stream's queue)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
keep = []
for n in range(10):
    e, ar, tr, cfg, shapes, out = bench.train_measure(torch, dev, 0, 1, None, "car_auto_T3", "car", 12, 6, 4, 2, True)
    print("streams created before: %d  ms/step %.3f" % (len(keep), e / 12 * 1e3), flush=True)
    keep.append(torch.cuda.Stream())
    del tr
