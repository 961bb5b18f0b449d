#!/bin/bash
# Static instruction count of the Ant's lane-group dynamics pass with ONE constraint row per lane instead of two (the
# per-lane row work of a 32-lanes-per-env form; its extra butterfly stage not included): compile the translation unit with
# -DGX_PROTO_ONE_ROW_PER_LANE and compare the step loop / Newton loop of group_dyn_tape_kernel<AntRobot, 1, 1, true>.
set -e
cd "$(dirname "$0")/../.."
GX_EXTRA_FLAGS_gx_kernels_ant="-DGX_PROTO_ONE_ROW_PER_LANE" python tools/build_variant.py ant_one_row > /dev/null
for lib in guardx_amd/lib/libguardx_hip.so guardx_amd/lib/variants/libguardx_hip_ant_one_row.so; do
  echo "== $lib"
  python tools/disasm_kernel.py "group_dyn_tape_kernel&AntRobotELi1ELi1ELb1" --lib $lib --loops
done
