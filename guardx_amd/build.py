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
SOURCES = ["gx_api.hip", "gx_kernels.hip", "gx_gae.hip", "gx_policy_step.hip", "gx_kernels_point.hip", "gx_kernels_point_bare.hip", "gx_kernels_swimmer.hip",
           "gx_kernels_point_split.hip", "gx_kernels_point_bare_split.hip", "gx_kernels_swimmer_split.hip",
           "gx_kernels_ant.hip", "gx_kernels_walker.hip"]
HEADERS = ["gx_device.h", "gx_robot.h", "gx_robot_ant.h", "gx_robot_ant_group.h", "gx_robot_legs.h", "gx_robot_legs_group.h", "gx_policy.h", "gx_kernels.h", "gx_robot_kernels.inl",
           "gx_split_rollout.inl", os.path.join("..", "..", "include", "guardx.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# No per-source flags.  History: LLVM's inter-procedural register allocation (on by default for
# amdgcn at -O3) lets a noinline callee use, without saving them, the VGPRs in whose lanes the CALLER parks spilled SGPRs
# (exec masks, v254/v255): after the call the masks are garbage and masked-off lanes store through garbage addresses
# (found with rocgdb in round 2 on group_rollout_kernel<WalkerRobot,...> with two call sites of the step; again in round
# 3 with ONE call site once the callee grew).  The cure that held: no call at all.  reset_done's fake step is tabulated
# with the layout pool (Pool::fake), so every kernel steps in one place, and both the Ant's and the Walker's lane-group
# steps are always_inline (gx_robot_ant_group.h, gx_robot_legs_group.h:substep_call); tests/test_native_abi.py asserts
# that no lane-group kernel contains an s_swappc_b64.
# Round 4: LLVM's "max-ilp" machine-scheduling strategy for the translation units that hold the two-kernel rollout of the
# robots whose dynamics pass is ONE wave per SIMD (the serial chains of the Point and the Swimmer; everything else of
# those robots -- step, reset, the closed-loop policy rollout, which lost 2.5 % with it -- keeps the default).  A lone wave issues a dependent vector instruction
# every ~5.75 cycles and an independent one every 4 (tools/probes/chain_clock_probe.hip); the default strategy schedules
# for occupancy / register pressure, this one interleaves independent instructions and halves the s_nop hazard fillers
# (538 -> 291 in the Point's dynamics pass).  Same instructions, same arithmetic, other order: results are bit-identical
# (the parity suite and the soaks run on it).  Same-box A/B: Point dynamics pass 91 -> 86.5 us, observation pass 33.6 ->
# 32.9 us, Swimmer dynamics pass 459 - 467 -> 451 us per 200 steps.  NOT for gx_kernels.hip: the layout sampler's phases are
# 2.4 % faster alone with it (554 -> 541 us) but the epoch, where they share every SIMD with other waves, is 2 % slower
# (0.517 -> 0.527 ms, 774 -> 759 M env-steps/s on one box); and not for the Ant's / Walker's lane-group kernels, which do
# not gain (1272 -> 1280, 2350 -> 2370 us).
_MAX_ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
# The Ant's lane-group step without the SLP vectorizer: its packed-fp32 forms (v_pk_mul_f32 / v_pk_add_f32, 340 in the
# step loop) cost more register shuffles than they save issue slots there -- 1231 -> 1152 us per 200 steps of 2000 envs
# (same-box A/B).  The Walker's is indifferent (2269 -> 2252 / 2274 us) and keeps the default; the Point's dynamics pass
# is 3 % SLOWER without it (87.0 -> 89.3 us).
PER_SOURCE_FLAGS = {"gx_kernels_point_split.hip": _MAX_ILP, "gx_kernels_point_bare_split.hip": _MAX_ILP,
                    "gx_kernels_swimmer_split.hip": _MAX_ILP, "gx_kernels_ant.hip": ["-fno-slp-vectorize"]}


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
    """`hipcc --version` in one line (HIP version + clang version): part of the build identity, because the kernels
    here are sensitive to what a particular compiler does (hipcc 7.2 miscompiled the CALL form of the Ant's / Walker's
    lane-group step -- the callee clobbered the caller's SGPR-spill VGPRs -- and is worked around by inlining it,
    gx_robot_legs_group.h; the soak and the parity suite were run on this compiler) -- a different compiler is a
    different build."""
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
