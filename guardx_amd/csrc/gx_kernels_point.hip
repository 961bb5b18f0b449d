// gx_kernels_point.hip -- instantiates the step / reset / rollout kernels for PointRobot
// (all but the two-kernel rollout: gx_kernels_point_split.hip).
#include "gx_robot_kernels.inl"

namespace gx {
GX_INSTANTIATE_REST(PointRobot)
} // namespace gx
