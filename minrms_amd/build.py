"""Builds libminrms_hip.so (the C-ABI engine) in-tree with hipcc for gfx950."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libminrms_hip.so")
SOURCES = ["minrms_engine.hip"]
HEADERS = ["dp_kernel.h", "fit_device.h", "layout.h", "nw_filter.h", "alt_solutions.h", "../../include/minrms_engine.h"]

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # float results must carry the reference's roundings: never fuse a*b+c
    "-ffp-contract=off",
    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
    "-Wno-unused-command-line-argument",
]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_engine(force=False, verbose=False):
    """Compile the engine; returns the path of the shared library."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + p.stdout)
    return LIB


if __name__ == "__main__":
    print(build_engine(force=True, verbose=True))
