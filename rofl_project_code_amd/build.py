"""Build librofl_zk.so (HIP, gfx950 only) in-tree with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "rofl_zk.hip")
DEPS = [os.path.join(HERE, "csrc", f) for f in ("rofl_zk.hip", "kernels.hpp", "fe32.hpp", "fe26.hpp", "keccak.hpp", "host51.hpp")] + [
    os.path.join(HERE, "..", "include", "rofl_zk.h")]
OUT = os.path.join(HERE, "librofl_zk.so")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-DROFL_FD_CHECK_HOST", "-std=c++17", "-shared", "-fPIC", "-o", OUT, SRC]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(OUT)
