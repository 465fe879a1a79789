"""Host time of Trainer.train_step per call (the stepping thread) and, from
events round every call, the device time of a step and the idle time between
steps on the step's stream -- with batches prebuilt / built by the loader
thread: is the step bound by its host, by gaps, or by slower kernels?"""
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
evs = []
def timed(self, *a, **k):
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    r = orig(self, *a, **k)
    acc['t'] += time.perf_counter() - t0
    acc['n'] += 1
    e1.record()
    evs.append((e0, e1))
    return r
train.Trainer.train_step = timed
for mode in ('prebuilt', 'thread', 'thread'):
    acc['t'] = 0.0; acc['n'] = 0
    el, ar, tr, cfg, shapes, out = bench.train_measure(torch, dev, 0, 1, None, 'car_auto_T3', 'car', 24, 8, 4, 2, mode)
    torch.cuda.synchronize()
    last = evs[-20:]
    busy = sum(a.elapsed_time(b) for a, b in last) / len(last)
    idle = sum(last[i][1].elapsed_time(last[i + 1][0])
               for i in range(len(last) - 1)) / (len(last) - 1)
    del evs[:]
    print(mode, 'ms/step %.3f' % (el / 24 * 1e3), 'host enqueue per train_step call %.3f ms' % (acc['t'] / acc['n'] * 1e3),
          '| device: step busy %.3f ms, idle between steps %.3f ms' % (busy, idle))
    del tr
