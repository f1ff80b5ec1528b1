from . import planner_utils, sdf_utils  # noqa: F401
