"""Build libguardx_hip.so (gfx950) in-tree with hipcc.

    python -m guardx_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so travels with the tree.
-ffp-contract=off: every fp32 operator in the kernels is one IEEE operation
(fused multiply-adds are written fmaf()), which is what makes the device
results reproducible against the CPU checker bit for bit.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libguardx_hip.so")
SOURCES = ["gx_api.hip", "gx_kernels.hip", "gx_gae.hip"]
HEADERS = ["gx_device.h", "gx_robot.h", "gx_robot_ant.h", "gx_policy.h", "gx_kernels.h", os.path.join("..", "..", "include", "guardx.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
