"""Builds tezip_amd/csrc/libtezip_hip.so for gfx950 with hipcc (in-tree, so the .so travels
to the GPU box with the repo snapshot).  hipcc cross-compiles without a GPU.

Flags that are part of the arithmetic contract (tz_math.hip.h): -ffp-contract=off and
correctly rounded float32 divide/sqrt."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["tz_api.hip", "tz_codec.hip", "tz_prednet.hip"]
HEADERS = ["tz_internal.h", "tz_math.hip.h", "tz_conv_kernels.hip.h", "tz_wino_kernels.hip.h", os.path.join("..", "..", "include", "tezip_hip.h")]
LIB = os.path.join(CSRC, "libtezip_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = (["-D" + d for d in os.environ.get("TEZIP_DEFINES", "").split()] if os.environ.get("TEZIP_DEFINES") else []) + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


STAMP = os.path.join(CSRC, ".build_flags")


def _flags_changed():
    """The objects in csrc/ were compiled with another flag set (TEZIP_DEFINES: the diagnostic variants of
    scripts/gpu_wino_ab.sh are built into the same libtezip_hip.so) -> everything is stale, whatever the mtimes say."""
    try:
        return open(STAMP).read() != " ".join(FLAGS)
    except OSError:
        return True


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    force = force or _flags_changed()
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            for err in ex.map(run, jobs):
                if verbose and err.strip():
                    print(err)
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs)
        with open(STAMP, "w") as f:
            f.write(" ".join(FLAGS))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
