// gx_kernels_swimmer.hip -- instantiates the step / reset / rollout kernels for SwimmerRobot.
#include "gx_robot_kernels.inl"

namespace gx {
template struct RobotLaunch<SwimmerRobot>;
} // namespace gx
