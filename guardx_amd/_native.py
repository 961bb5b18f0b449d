"""ctypes binding of libguardx_hip.so (include/guardx.h).

There is no CPU fallback: if the HIP library is missing or fails to load this
module raises, and every Engine call goes through the C ABI below.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libguardx_hip.so")

GX_OK, GX_ERR_ARG, GX_ERR_UNSUPPORTED, GX_ERR_LAYOUT, GX_ERR_HIP, GX_ERR_STATE = range(6)


class GxConfig(C.Structure):
    """Mirror of `struct gx_config` (include/guardx.h)."""
    _fields_ = [
        ("struct_size", C.c_int32), ("robot", C.c_int32), ("env_num", C.c_int32),
        ("env_total", C.c_int32), ("env_offset", C.c_int32), ("seed", C.c_uint32),
        ("num_steps", C.c_int32), ("hazards_num", C.c_int32), ("lidar_num_bins", C.c_int32),
        ("lidar_alias", C.c_int32), ("lidar_max_dist_set", C.c_int32),
        ("lidar_max_dist", C.c_float), ("lidar_exp_gain", C.c_float),
        ("goal_size", C.c_float), ("hazards_size", C.c_float), ("reward_distance", C.c_float),
        ("goal_keepout", C.c_double), ("hazards_keepout", C.c_double),
        ("robot_keepout", C.c_double), ("placements_margin", C.c_double),
        ("extents", C.c_double * 4),
        ("observe_goal_lidar", C.c_int32), ("observe_goal_comp", C.c_int32),
        ("observe_hazards", C.c_int32), ("observe_qpos", C.c_int32),
        ("observe_qvel", C.c_int32), ("observe_ctrl", C.c_int32),
        ("observe_vel", C.c_int32), ("observe_acc", C.c_int32),
        ("n_candidates", C.c_int32), ("physics_steps", C.c_int32),
        ("robot_goal_min_dist", C.c_float), ("device", C.c_int32),
        ("placements", C.POINTER(C.c_double)),
        ("pillars_num", C.c_int32), ("observe_pillars", C.c_int32), ("pillars_size", C.c_float),
        ("robot_rot", C.c_float), ("pillars_keepout", C.c_double),
    ]


class GxPolicy(C.Structure):
    """Mirror of `struct gx_policy`."""
    _fields_ = [("struct_size", C.c_int32), ("hidden", C.c_int32), ("d_params", C.c_void_p),
                ("seed", C.c_uint32 * 2)]


# every symbol include/guardx.h declares: name -> (restype, argtypes)
_FP = C.c_void_p  # device pointers travel as integers
_HFP = C.POINTER(C.c_float)
_U32P = C.POINTER(C.c_uint32)
_I32P = C.POINTER(C.c_int32)
SYMBOLS = {
    "gx_last_error": (C.c_char_p, []),
    "gx_abi_version": (C.c_int32, []),
    "gx_build_id": (C.c_char_p, []),
    "gx_build_compiler": (C.c_char_p, []),
    "gx_create": (C.c_int, [C.POINTER(GxConfig), C.POINTER(C.c_void_p)]),
    "gx_destroy": (C.c_int, [C.c_void_p]),
    "gx_obs_dim": (C.c_int32, [C.c_void_p]),
    "gx_act_dim": (C.c_int32, [C.c_void_p]),
    "gx_dims": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int32)] * 4),
    "gx_reset": (C.c_int, [C.c_void_p, _FP, C.c_void_p]),
    "gx_layout_size": (C.c_int, [C.c_void_p, _I32P]),
    "gx_layout_size_min": (C.c_int, [C.c_void_p, _I32P]),
    "gx_step": (C.c_int, [C.c_void_p, _FP, _FP, _FP, _FP, _FP, _FP, C.c_void_p]),
    "gx_step_rd": (C.c_int, [C.c_void_p, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _I32P, C.c_void_p]),
    "gx_step_set_floats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "gx_step_slab": (C.c_int, [C.c_void_p, _FP, _FP, C.c_int32, C.c_int32, _I32P, C.c_void_p]),
    "gx_reset_done_commit": (C.c_int, [C.c_void_p]),
    "gx_reset_done": (C.c_int, [C.c_void_p, _FP, _FP, C.c_void_p]),
    "gx_rollout_packed": (C.c_int, [C.c_void_p, C.c_int32, _FP, _FP, C.c_void_p]),
    "gx_packed_width": (C.c_int32, [C.c_void_p]),
    "gx_tape_floats": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gx_rollout_tape": (C.c_int, [C.c_void_p, C.c_int32, _FP, _FP, C.POINTER(C.c_int64), C.c_void_p]),
    "gx_expand_tape": (C.c_int, [C.c_void_p, C.c_int32, _FP, C.c_int64, _FP, C.c_void_p]),
    "gx_expand_tapes": (C.c_int, [C.c_void_p, C.c_int32, _FP, C.c_int64, C.c_int32, C.c_int64, _FP, C.c_int64, C.c_void_p]),
    "gx_sample_shard": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _FP, C.c_int32, _FP, C.c_void_p]),
    "gx_reset_from_shards": (C.c_int, [C.c_void_p, _FP, _FP, C.c_int32, C.c_int32, _FP, C.c_void_p]),
    "gx_set_layout_source": (C.c_int, [C.c_void_p, C.c_int32]),
    "gx_aux_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "gx_aux_stream_renew": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "gx_shard_block_floats": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    "gx_sample_shard_ahead": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _FP, C.c_int32, C.POINTER(C.c_int64), C.c_void_p]),
    "gx_shard_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gx_install_shards": (C.c_int, [C.c_void_p, C.c_int64, _FP, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "gx_rollout": (C.c_int, [C.c_void_p, C.c_int32, _FP, _FP, _FP, _FP, _FP, C.c_void_p]),
    "gx_rollout_policy": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(GxPolicy)] + [_FP] * 12 + [C.c_void_p]),
    "gx_set_policy_impl": (C.c_int, [C.c_void_p, C.c_int32]),
    "gx_math_probe2": (C.c_int, [C.c_int32, _FP, _FP, _FP, C.c_void_p]),
    "gx_get_state": (C.c_int, [C.c_void_p] + [_HFP] * 8 + [_U32P, _I32P]),
    "gx_set_state": (C.c_int, [C.c_void_p] + [_HFP] * 8 + [_U32P, _I32P]),
    "gx_get_pool": (C.c_int, [C.c_void_p, _HFP, C.c_int32, _I32P]),
    "gx_set_prefetch": (C.c_int, [C.c_void_p, C.c_int32]),
    "gx_prefetch_stats": (C.c_int, [C.c_void_p, _I32P, _I32P, _I32P]),
    "gx_set_path": (C.c_int, [C.c_void_p, C.c_int32]),
    "gx_debug_stamps": (C.c_int, [C.c_void_p, _FP]),
    "gx_buffer_store": (C.c_int, [C.c_int32] * 5 + [_FP] * 14 + [C.c_void_p]),
    "gx_gae_finish_path": (C.c_int, [C.c_int32] * 3 + [_FP] * 5 + [C.c_double, C.c_double, _FP, _FP, C.c_int32, C.c_void_p]),
    "gx_gae_rollout": (C.c_int, [C.c_int32, C.c_int32, _FP, _FP, _FP, _FP, C.c_double, C.c_double, _FP, _FP, C.c_void_p]),
    "gx_adv_normalize": (C.c_int, [C.c_int32, C.c_int32, _FP, C.c_int32, C.c_void_p]),
    "gx_math_probe": (C.c_int, [C.c_int32, _FP, _FP, _FP, _FP, _FP, _FP, C.c_void_p]),
    "gx_split_probe": (C.c_int, [_U32P, C.c_int32, _FP, C.c_void_p]),
}

_lib = None


def load():
    """Load the HIP library; raises (never falls back) when it is unavailable or was built from other sources.

    The library carries the hash of the sources it was built from (gx_build_id).  If the file is missing or the
    hash on disk differs from the sources in the tree it is rebuilt in place with hipcc (one process at a time:
    guardx_amd.build takes a file lock and renames the finished file into place); a library whose embedded id does
    not match after that is refused."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build as _build
    want = _build.source_hash()
    if _build.needs_build():
        try:
            _build.build(force=False)
        except Exception as exc:  # noqa: BLE001
            raise ImportError(
                f"{LIB_PATH} is missing or stale (sources {want}, library {_build.built_id()}) and could not be "
                f"built with hipcc ({exc}); run `python -m guardx_amd.build` (guardx_amd has no CPU fallback)") from exc
    path = LIB_PATH
    if os.environ.get("GX_LIB") and os.environ.get("GX_LIB_EXPERIMENT") == "1":
        # A/B experiments only (tools/build_variant.py): a variant built from the SAME sources with other flags
        path = os.path.abspath(os.environ["GX_LIB"])
        import warnings
        warnings.warn(f"guardx_amd: loading the EXPERIMENTAL library {path} (GX_LIB + GX_LIB_EXPERIMENT=1); its build id is not "
                      "checked against the tree", RuntimeWarning, stacklevel=3)
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted
        fn.restype = res
        fn.argtypes = args
    got = lib.gx_build_id().decode()
    if got != want and path == LIB_PATH:
        raise ImportError(f"{LIB_PATH} was built from other sources (library {got}, tree {want}); "
                          "run `python -m guardx_amd.build`")
    _warn_if_unprofiled_compiler(lib)
    _lib = lib
    return lib


def _warn_if_unprofiled_compiler(lib):
    """The parity soak and the committed profiles were taken on one hipcc (profiles/<round>_build_id.txt, line 2); the
    lane-group kernels are known to be sensitive to the compiler (guardx_amd/build.py).  A library built by another one
    is not refused -- the parity tests are the judge -- but it says so once."""
    import glob
    import warnings
    ids = sorted(glob.glob(os.path.join(os.path.dirname(_HERE), "profiles", "r[0-9][0-9]_build_id.txt")))
    if not ids:
        return
    try:
        lines = open(ids[-1]).read().splitlines()
    except OSError:
        return
    have = lib.gx_build_compiler().decode()
    if len(lines) > 1 and lines[1].strip() and have != "unknown" and lines[1].strip() != have:
        warnings.warn(f"libguardx_hip.so was built with [{have}]; the parity soak and profiles of {os.path.basename(ids[-1])} "
                      f"were taken with [{lines[1].strip()}]: run `pytest -m gpu` and tests/soak_parity.py on this build",
                      RuntimeWarning, stacklevel=3)


class GxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"guardx status {status}: {msg}")
        self.status = status


def check(status):
    if status != GX_OK:
        msg = load().gx_last_error()
        raise GxError(status, msg.decode() if msg else "")
