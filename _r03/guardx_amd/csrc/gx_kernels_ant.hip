// gx_kernels_ant.hip -- instantiates the step / reset / rollout kernels for AntRobot.
#include "gx_robot_kernels.inl"

namespace gx {
template struct RobotLaunch<AntRobot>;
} // namespace gx
