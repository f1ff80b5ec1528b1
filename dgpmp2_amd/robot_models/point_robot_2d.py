from .robot_model import RobotModel


class PointRobot2D(RobotModel):
  """2-D point robot, state [x, y, vx, vy] (reference: robot_models/point_robot_2d.py:5-11).  `batch_size` and
  `num_traj_states` are accepted for signature compatibility; nothing is pre-sized here."""

  def __init__(self, sphere_radii, batch_size=1, num_traj_states=1, use_cuda=False):
    super(PointRobot2D, self).__init__(2, 1, 2, 4, sphere_radii, batch_size, num_traj_states, use_cuda)
