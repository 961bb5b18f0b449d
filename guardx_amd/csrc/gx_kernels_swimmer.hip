// gx_kernels_swimmer.hip -- instantiates the step / reset / rollout kernels for SwimmerRobot
// (all but the two-kernel rollout: gx_kernels_swimmer_split.hip).
#include "gx_robot_kernels.inl"

namespace gx {
GX_INSTANTIATE_REST(SwimmerRobot)
} // namespace gx
