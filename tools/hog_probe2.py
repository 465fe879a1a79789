#!/usr/bin/env python
"""Can a 248-workgroup persistent kernel (133 KB LDS each) start at once while
8 other workgroups (96 KB LDS, 1024 threads: a single-launch kd-tree build)
already sit on 8 CUs?  Repeated with different lead times; the persistent
kernel should take its own spin time (1 ms), not twice that."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pointgnn_amd.engine import concurrent_streams  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    lib = ctypes.CDLL(os.path.join(ROOT, "ab", "libcuprobe.so"))
    VP, I = ctypes.c_void_p, ctypes.c_int
    lib.hog_launch.argtypes = [VP, I, I, I, I, VP]
    dev = torch.device("cuda")
    sa, sb = concurrent_streams(2)
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    for small_wg, small_lds, small_thr in ((8, 98304, 1024), (1, 98304, 1024),
                                           (8, 32768, 256), (32, 98304, 1024)):
        for big_wg in (248, 256):
            res = []
            for rep in range(12):
                torch.cuda.synchronize()
                lib.hog_launch(None, small_wg, small_thr, small_lds, 1500,
                               sb.cuda_stream)
                time.sleep(0.0001 * (rep % 4))
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(sa):
                    e0.record()
                    lib.hog_launch(None, big_wg, 512, 136640, 1000,
                                   sa.cuda_stream)
                    e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1))
            print("%2d wg x %4d thr x %6d B resident for 1.5 ms, then %d wg x 133 KB "
                  "spinning 1 ms: %s" % (small_wg, small_thr, small_lds, big_wg,
                                         " ".join("%.2f" % r for r in res)))


if __name__ == "__main__":
    main()
