// gx_kernels_point.hip -- instantiates the step / reset / rollout kernels for PointRobot.
#include "gx_robot_kernels.inl"

namespace gx {
template struct RobotLaunch<PointRobot>;
} // namespace gx
