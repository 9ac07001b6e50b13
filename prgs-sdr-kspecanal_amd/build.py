"""Build libksa.so (HIP kernels + C ABI) for gfx950 with hipcc.  In-tree, no JIT cache:
the .so sits next to this file so that it travels to the GPU box with the repo snapshot."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "ksa_api.hip")
OUT = os.path.join(HERE, "libksa.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Wno-unused-value",
         "-shared", "-fPIC"]


def sources():
    d = os.path.join(HERE, "csrc")
    return [os.path.join(d, f) for f in sorted(os.listdir(d))] + [os.path.join(HERE, "..", "include", "ksa.h")]


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force=False, verbose=False):
    """Compile if the library is missing or older than its sources.  Returns the .so path."""
    if not force and not is_stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-o", OUT + ".tmp", SRC]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
