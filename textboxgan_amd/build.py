"""Build libtbg_hip.so (gfx950) in-tree with hipcc.  The role of the reference's nvcc JIT
(custom_ops.py:109-213) -- but ahead of time, one shared library, C ABI, no framework headers."""
from __future__ import annotations

import fcntl
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtbg_hip.so")
HOST_LIB = os.path.join(HERE, "libtbg_host.so")
SOURCES = ["elementwise.hip", "upfirdn.hip", "conv.hip", "conv_units.hip", "conv_units_s2.hip", "conv_small.hip", "rgb.hip", "lstm.hip", "smalls.hip", "host_util.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _digest() -> str:
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ["../../include/tbg.h"]:
        path = os.path.join(CSRC, name)
        if os.path.isfile(path):
            with open(path, "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build_native(force: bool = False, verbose: bool = True) -> str:
    stamp = LIB + ".sha256"
    dig = _digest()

    def fresh():
        return (os.path.exists(LIB) and os.path.exists(HOST_LIB) and os.path.exists(stamp)
                and open(stamp).read().strip() == dig)

    if not force and fresh():
        return LIB
    objdir = os.path.join(CSRC, ".obj")
    os.makedirs(objdir, exist_ok=True)
    # one builder at a time (every rank of a multi-process launch calls this): the others wait here and then find the
    # library fresh.  Objects and libraries are written to "<name>.tmp.<pid>" and renamed on success, so that a killed
    # compile can never leave a truncated file under a final name (the role of custom_ops.py's atomic rename, :196-206).
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and fresh():
            return LIB
        _build_locked(force, verbose, objdir)
        with open(stamp + f".tmp.{os.getpid()}", "w") as f:
            f.write(dig)
        os.replace(stamp + f".tmp.{os.getpid()}", stamp)
    return LIB


def _build_locked(force: bool, verbose: bool, objdir: str) -> None:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    pid = os.getpid()
    # one object per translation unit, compiled in parallel and cached by content (sources + headers + flags) under
    # csrc/.obj (git-ignored): editing one kernel file recompiles that file only
    hdr = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ["../../include/tbg.h"]:
        if name.endswith(".h"):
            with open(os.path.join(CSRC, name), "rb") as f:
                hdr.update(name.encode() + b"\0" + f.read())
    hdr.update(" ".join(FLAGS).encode())
    jobs, objs = [], []
    for src in SOURCES:
        with open(os.path.join(CSRC, src), "rb") as f:
            key = hashlib.sha256(hdr.digest() + f.read()).hexdigest()[:16]
        obj = os.path.join(objdir, f"{src}.{key}.o")
        objs.append(obj)
        if force or not os.path.exists(obj):
            for old in os.listdir(objdir):
                if old.startswith(src + "."):
                    os.remove(os.path.join(objdir, old))
            tmp = f"{obj}.tmp.{pid}"
            cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", tmp]
            if verbose:
                print("[tbg build]", " ".join(cmd), flush=True)
            jobs.append((src, tmp, obj, subprocess.Popen(cmd)))
    failed = None
    for src, tmp, obj, job in jobs:  # every job is waited for, also after a failure (no orphaned compilers)
        if job.wait() == 0 and failed is None:
            os.replace(tmp, obj)
        else:
            failed = failed or (job.returncode, src)
            if os.path.exists(tmp):
                os.remove(tmp)
    if failed is not None:
        raise subprocess.CalledProcessError(failed[0], f"hipcc -c {failed[1]}")
    tmp = f"{LIB}.tmp.{pid}"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs]
    if verbose:
        print("[tbg build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    # the host-only helpers once more WITHOUT the HIP runtime (g++): the checkpoint checksum on CPU-only hosts
    tmp = f"{HOST_LIB}.tmp.{pid}"
    host_cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "c++",
                os.path.join(CSRC, "host_util.hip"), "-o", tmp]
    if verbose:
        print("[tbg build]", " ".join(host_cmd), flush=True)
    subprocess.check_call(host_cmd)
    os.replace(tmp, HOST_LIB)


if __name__ == "__main__":
    build_native(force="--force" in sys.argv)
    print(LIB)
