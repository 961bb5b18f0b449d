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
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB = os.path.join(LIB_DIR, "libguardx_hip.so")
# one translation unit per robot (gx_robot_kernels.inl instantiated for it), compiled in parallel
SOURCES = ["gx_api.hip", "gx_kernels.hip", "gx_gae.hip", "gx_kernels_point.hip", "gx_kernels_point_bare.hip", "gx_kernels_swimmer.hip",
           "gx_kernels_ant.hip", "gx_kernels_walker.hip"]
HEADERS = ["gx_device.h", "gx_robot.h", "gx_robot_ant.h", "gx_robot_ant_group.h", "gx_robot_legs.h", "gx_robot_legs_group.h", "gx_policy.h", "gx_kernels.h", "gx_robot_kernels.inl",
           os.path.join("..", "..", "include", "guardx.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# The Ant / Walker steps are real (noinline) device functions called from the lane-group kernels.  With LLVM's
# inter-procedural register allocation (on by default for amdgcn at -O3) hipcc 7.2 lets such a callee use, without
# saving them, the VGPRs in whose lanes the CALLER parks spilled SGPRs (exec masks, v254/v255 in the failing
# kernel): after the call the masks are garbage and masked-off lanes execute stores with garbage addresses
# (found with rocgdb on group_rollout_kernel<WalkerRobot,5,4,...>: HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in
# the observation-row stores right after the second substep_call; gone with IPRA off).  So those two translation
# units are compiled without IPRA: the callee then saves and restores the callee-saved registers it uses.
PER_SOURCE_FLAGS = {
    "gx_kernels_ant.hip": ["-mllvm", "-enable-ipra=0"],
    "gx_kernels_walker.hip": ["-mllvm", "-enable-ipra=0"],
}


def _extra(src):
    # GX_EXTRA_FLAGS_<source stem>="...": experiments (replaces the per-source defaults when it starts with "=")
    env = os.environ.get("GX_EXTRA_FLAGS_" + os.path.splitext(src)[0], "")
    if env.startswith("="):
        return env[1:].split()
    return PER_SOURCE_FLAGS.get(src, []) + env.split()


def _deps_mtime():
    deps = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    return max(os.path.getmtime(d) for d in deps if os.path.exists(d))


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return _deps_mtime() > t or any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


def build(force=False, verbose=False, jobs=None):
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    hdr_t = _deps_mtime()

    def compile_one(src):
        path, obj = os.path.join(CSRC, src), _obj(src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_t):
            return
        cmd = [hipcc] + FLAGS + _extra(src) + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    jobs = jobs or int(os.environ.get("GX_BUILD_JOBS", "0")) or min(len(SOURCES), os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + [_obj(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
