"""Host time of Trainer.train_step per call (the stepping thread) with batches
prebuilt / built by the loader thread: is the step bound by its host?"""
import sys, time, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import pointgnn_amd
from pointgnn_amd import train
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
orig = train.Trainer.train_step
acc = {'t': 0.0, 'n': 0}
def timed(self, *a, **k):
    t0 = time.perf_counter()
    r = orig(self, *a, **k)
    acc['t'] += time.perf_counter() - t0
    acc['n'] += 1
    return r
train.Trainer.train_step = timed
for mode in ('prebuilt', 'thread', 'thread'):
    acc['t'] = 0.0; acc['n'] = 0
    el, ar, tr, cfg, shapes, out = bench.train_measure(torch, dev, 0, 1, None, 'car_auto_T3', 'car', 24, 8, 4, 2, mode)
    print(mode, 'ms/step %.3f' % (el / 24 * 1e3), 'host enqueue per train_step call %.3f ms' % (acc['t'] / acc['n'] * 1e3))
    del tr
