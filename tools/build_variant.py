#!/usr/bin/env python3
"""Build an experimental variant of libguardx_hip.so beside the product library, for same-box A/B runs:

    GX_EXTRA_FLAGS_gx_kernels_swimmer_split="=" python tools/build_variant.py noilp
    GX_LIB=guardx_amd/lib/variants/libguardx_hip_noilp.so python tools/ab_epoch.py Goal_Swimmer_8Hazards

Objects and library go to guardx_amd/lib/variants/ (git-ignored like the rest of lib/); the product library, its objects
and BUILD_ID are not touched.  GX_LIB is honoured by guardx_amd._native only together with GX_LIB_EXPERIMENT=1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from guardx_amd import build as b  # noqa: E402

tag = sys.argv[1]
vdir = os.path.join(b.LIB_DIR, "variants")
b.OBJ_DIR = os.path.join(vdir, "obj_" + tag)
b.LIB = os.path.join(vdir, f"libguardx_hip_{tag}.so")
b.BUILD_ID_FILE = os.path.join(vdir, f"BUILD_ID_{tag}")
os.makedirs(b.OBJ_DIR, exist_ok=True)
print(b._build_locked(False, "-v" in sys.argv, None))
