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
           "gx_split_rollout.inl", os.path.join("..", "..", "include", "guardx.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# No per-source flags any more.  History: LLVM's inter-procedural register allocation (on by default for amdgcn at
# -O3) let a noinline callee use, without saving them, the VGPRs in whose lanes the CALLER parks spilled SGPRs (exec
# masks, v254/v255): after the call the masks are garbage and masked-off lanes store through garbage addresses (found
# with rocgdb on group_rollout_kernel<WalkerRobot,5,4,...> when it had TWO call sites of the step:
# HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in the observation-row stores right after the second substep_call), and
# the Ant / Walker translation units were built with -mllvm -enable-ipra=0.  Since reset_done's fake step is tabulated
# with the layout pool (Pool::fake) every kernel has ONE call site of the step: the Ant's steps are inlined (no call
# at all); the Walker's lane-group step stays a call (inlining it miscompiles, gx_robot_legs_group.h) and is built
# with IPRA again -- every parity test and the soak pass, and it is 18 % faster than with the callee saving its
# callee-saved registers.
PER_SOURCE_FLAGS = {}


def _extra(src):
    # GX_EXTRA_FLAGS_<source stem>="...": experiments (replaces the per-source defaults when it starts with "=")
    env = os.environ.get("GX_EXTRA_FLAGS_" + os.path.splitext(src)[0], "")
    if env.startswith("="):
        return env[1:].split()
    return PER_SOURCE_FLAGS.get(src, []) + env.split()


BUILD_ID_FILE = os.path.join(LIB_DIR, "BUILD_ID")
LOCK_FILE = os.path.join(LIB_DIR, ".build.lock")


_COMPILER = None


def compiler_id():
    """`hipcc --version` in one line (HIP version + clang version): part of the build identity, because at least one
    kernel here depends on what a particular compiler does (the Walker's lane-group step must stay a call:
    inlining it miscompiles, gx_robot_legs_group.h) -- a different compiler is a different build."""
    global _COMPILER
    if _COMPILER is None:
        try:
            out = subprocess.run([os.environ.get("HIPCC", "hipcc"), "--version"], capture_output=True, text=True,
                                 timeout=60).stdout
            keep = [ln.strip() for ln in out.splitlines() if ln.startswith(("HIP version", "AMD clang version"))]
            _COMPILER = "; ".join(keep) or "unknown"
        except Exception:  # noqa: BLE001 - no compiler on this machine: the prebuilt library's own record stands
            _COMPILER = "unknown"
    return _COMPILER


def source_hash():
    """sha256 over every source, header, flag and the compiler version that goes into the library: the identity of
    a build.  It is compiled into the library (gx_build_id()) and checked at load time, so a stale or foreign .so
    is never loaded silently, whatever the file times say (the tree is copied to the GPU box without them)."""
    import hashlib
    h = hashlib.sha256()
    h.update(compiler_id().encode() + b"\0")
    names = sorted(set(SOURCES) | set(HEADERS) | {"gx_split_rollout.inl"})
    for n in names:
        path = os.path.join(CSRC, n)
        h.update(n.encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(repr((FLAGS, sorted(PER_SOURCE_FLAGS.items()))).encode())
    return h.hexdigest()[:24]


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")


def built_id():
    try:
        with open(BUILD_ID_FILE) as f:
            return f.read().strip()
    except OSError:
        return None


def needs_build():
    return not os.path.exists(LIB) or built_id() != source_hash()


def _dep_hash(src):
    """identity of one object file: its source, every header, its flags (objects are reused across builds)"""
    import hashlib
    h = hashlib.sha256()
    for n in [src] + sorted(set(HEADERS) | {"gx_split_rollout.inl"}):
        with open(os.path.join(CSRC, n), "rb") as f:
            h.update(n.encode() + b"\0" + f.read())
    h.update(repr((FLAGS, _extra(src), compiler_id())).encode())
    return h.hexdigest()[:24]


def build(force=False, verbose=False, jobs=None):
    """Build under an inter-process lock (several ranks importing at once build once), link to a temporary name
    and rename into place (nobody can dlopen a half-written file)."""
    import fcntl
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(LOCK_FILE, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB
            return _build_locked(force, verbose, jobs)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose, jobs):
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "hipcc")
    bid = source_hash()

    def compile_one(src):
        path, obj, tag = os.path.join(CSRC, src), _obj(src), _obj(src) + ".id"
        want = _dep_hash(src) + (":" + bid if src == "gx_api.hip" else "")   # gx_api.hip carries the build id
        try:
            have = open(tag).read().strip()
        except OSError:
            have = None
        if not force and os.path.exists(obj) and have == want:
            return
        cmd = [hipcc] + FLAGS + _extra(src) + \
              (['-DGX_BUILD_ID="%s"' % bid, '-DGX_BUILD_COMPILER="%s"' % compiler_id().replace('"', "'")]
               if src == "gx_api.hip" else []) + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(tag, "w") as f:
            f.write(want)

    jobs = jobs or int(os.environ.get("GX_BUILD_JOBS", "0")) or min(len(SOURCES), os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(compile_one, SOURCES))
    tmp = LIB + ".tmp.%d" % os.getpid()
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", tmp] + [_obj(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    with open(BUILD_ID_FILE + ".tmp", "w") as f:
        f.write(bid + "\n")
    os.replace(BUILD_ID_FILE + ".tmp", BUILD_ID_FILE)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print("build id", built_id())
