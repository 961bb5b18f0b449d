// gx_kernels_point_bare.hip -- the same kernels for the round-1 reading of point.xml's actuators
// (general actuators without the class defaults, robot id 4; see gx_robot.h).
#include "gx_robot_kernels.inl"

namespace gx {
template struct RobotLaunch<PointBareRobot>;
} // namespace gx
