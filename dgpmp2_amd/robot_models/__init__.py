from .robot_model import RobotModel
from .point_robot_2d import PointRobot2D
from .point_robot_xyh import PointRobotXYH
