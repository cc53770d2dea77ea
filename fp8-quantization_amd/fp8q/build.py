"""Compile csrc/*.hip -> csrc/libfp8q_hip.so for gfx950 (hipcc cross-compiles, no GPU needed).

Six translation units (quantize / min-max family, MSE grid, fused epilogue, storage codes, the float64 lane, the interval-histogram MSE search) compiled in parallel and
linked into one library.  In-tree build: the .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import concurrent.futures
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
SO = os.path.join(CSRC, "libfp8q_hip.so")
SOURCES = ["fp8q_quant.hip", "fp8q_mse.hip", "fp8q_epilogue.hip", "fp8q_codec.hip", "fp8q_f64.hip", "fp8q_mse_hist.hip"]
HEADERS = ["fp8q_common.h", "fp8q_device.h", "fp8q_tables.h", os.path.join("..", "..", "include", "fp8q.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]


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
    hipcc = _hipcc()
    with tempfile.TemporaryDirectory(prefix="fp8q_build_") as tmp:
        def compile_one(src):
            obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
            cmd = [hipcc] + CFLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd, cwd=CSRC)
            return obj
        with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", SO + ".tmp"]
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link, cwd=CSRC)
    os.replace(SO + ".tmp", SO)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
