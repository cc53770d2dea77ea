"""Compile csrc/fp8q_kernels.hip -> csrc/libfp8q_hip.so for gfx950 (hipcc cross-compiles, no GPU needed).

In-tree build: the .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
SO = os.path.join(CSRC, "libfp8q_hip.so")
SOURCES = ["fp8q_kernels.hip"]
HEADERS = ["fp8q_device.h", "fp8q_tables.h", os.path.join("..", "..", "include", "fp8q.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    cmd = [_hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", SO + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(SO + ".tmp", SO)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
