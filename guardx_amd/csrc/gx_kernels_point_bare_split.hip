// gx_kernels_point_bare_split.hip -- the two-kernel rollout (gx_split_rollout.inl: dynamics tape + observation pass) of
// PointBareRobot, in a translation unit of its own: its dynamics pass is ONE wave per SIMD, and guardx_amd/build.py compiles this
// unit with LLVM's max-ilp scheduling strategy (see PER_SOURCE_FLAGS there).
#include "gx_robot_kernels.inl"

namespace gx {
GX_INSTANTIATE_SPLIT(PointBareRobot)
} // namespace gx
