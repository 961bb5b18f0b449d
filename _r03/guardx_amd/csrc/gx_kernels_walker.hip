// gx_kernels_walker.hip -- instantiates the step / reset / rollout kernels for WalkerRobot.
#include "gx_robot_kernels.inl"

namespace gx {
template struct RobotLaunch<WalkerRobot>;
} // namespace gx
