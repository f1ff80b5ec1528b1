from .robot_model import RobotModel


class PointRobotXYH(RobotModel):
  """(x, y, theta) point robot, state [x, y, th, vx, vy, w] (reference: robot_models/point_robot_xyh.py:5-11)."""

  def __init__(self, sphere_radii, use_cuda=False, batch_size=1, num_traj_states=1):
    super(PointRobotXYH, self).__init__(3, 1, 2, 6, sphere_radii, batch_size, num_traj_states, use_cuda)
