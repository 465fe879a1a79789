"""Builds libpointgnn_hip.so (gfx950) in-tree with hipcc.

`python -m pointgnn_amd.build` or `pointgnn_amd.build.build()`.  hipcc
cross-compiles for gfx950 without a GPU.  Objects go to csrc/build/, the
shared library next to this file (git-ignored; it travels to the GPU box with
the working tree).
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(_ROOT, "include")
LIB_PATH = os.path.join(_HERE, "libpointgnn_hip.so")
ARCH = "gfx950"

CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH,
            "-ffp-contract=off",  # radius predicate must not fuse mul+add
            "-Wall", "-Wno-unused-function",
            "-I", INCLUDE, "-I", CSRC]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libpointgnn_hip.so")


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                  if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)
           if f.endswith(".h")]
    return sorted(hs)


def _stamp(paths, extra=""):
    h = hashlib.sha1(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=True, extra_flags=()):
    """Compile every csrc/*.hip for gfx950 and link the shared library.
    Incremental: an object is rebuilt when its source, any header or the flags
    changed."""
    hipcc = _hipcc()
    bdir = os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    flags = CXXFLAGS + list(extra_flags)
    hdr_stamp = _stamp(_headers(), " ".join(flags))
    jobs = []
    objs = []
    for src in _sources():
        obj = os.path.join(bdir, os.path.basename(src)[:-4] + ".o")
        stamp_file = obj + ".stamp"
        stamp = _stamp([src], hdr_stamp)
        objs.append(obj)
        old = open(stamp_file).read() if os.path.exists(stamp_file) else ""
        if force or old != stamp or not os.path.exists(obj):
            jobs.append((src, obj, stamp_file, stamp))

    def compile_one(job):
        src, obj, stamp_file, stamp = job
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print("[pointgnn_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(stamp_file, "w") as f:
            f.write(stamp)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    need_link = bool(jobs) or not os.path.exists(LIB_PATH)
    if need_link:
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o",
               LIB_PATH] + objs
        if verbose:
            print("[pointgnn_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
