// gx_kernels_point_bare.hip -- the same kernels for the round-1 reading of point.xml's actuators
// (general actuators without the class defaults, robot id 4; see gx_robot.h)
// (all but the two-kernel rollout: gx_kernels_point_bare_split.hip).
#include "gx_robot_kernels.inl"

namespace gx {
GX_INSTANTIATE_REST(PointBareRobot)
} // namespace gx
