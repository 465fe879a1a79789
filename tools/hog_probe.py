#!/usr/bin/env python
"""Where does a small kernel land, and how long does it wait, while a
persistent kernel holds most of every CU's LDS?  (tools/micro/cu_probe.hip)"""
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pointgnn_amd.engine import concurrent_streams  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def places(t, n):
    v = t.view(-1, 2)[:n].cpu().numpy().astype("uint32")
    xcc, hw = v[:, 0] & 0xF, v[:, 1]
    return collections.Counter(zip(xcc.tolist(), ((hw >> 13) & 7).tolist(),
                                   ((hw >> 8) & 0xF).tolist()))


def main():
    lib = ctypes.CDLL(os.path.join(ROOT, "ab", "libcuprobe.so"))
    VP, I = ctypes.c_void_p, ctypes.c_int
    lib.hog_launch.argtypes = [VP, I, I, I, I, VP]
    lib.cu_probe_lds_launch.argtypes = [VP, I, I, I, I, VP]
    dev = torch.device("cuda")
    sa, sb = concurrent_streams(2)
    hog_out = torch.zeros(2 * 512, dtype=torch.int32, device=dev)
    pr_out = torch.zeros(2 * 4096, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for hog_blocks in (256, 248, 240):
        for hog_threads in (512,):
            for (pb, pt, plds) in ((1, 1024, 98304), (8, 256, 32768),
                                   (64, 256, 32768), (64, 256, 0),
                                   (1024, 256, 32768)):
                hog_out.zero_()
                pr_out.zero_()
                torch.cuda.synchronize()
                lib.hog_launch(hog_out.data_ptr(), hog_blocks, hog_threads,
                               136640, 2000, sa.cuda_stream)
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                import time
                time.sleep(0.0005)
                with torch.cuda.stream(sb):
                    e0.record()
                    lib.cu_probe_lds_launch(pr_out.data_ptr(), pb, pt, 2000,
                                            plds, sb.cuda_stream)
                    e1.record()
                torch.cuda.synchronize()
                hp = places(hog_out, hog_blocks)
                pp = places(pr_out, pb)
                per_xcc = collections.Counter(k[0] for k in hp)
                print("hog %d wg x %d thr (133 KB): CUs %d per XCC %s | probe %4d wg x %4d thr "
                      "lds %6d: %.3f ms, CUs %d, shared with hog %d" % (
                          hog_blocks, hog_threads, len(hp),
                          [per_xcc[i] for i in range(8)], pb, pt, plds,
                          e0.elapsed_time(e1), len(pp),
                          len(set(pp) & set(hp))))


if __name__ == "__main__":
    main()
