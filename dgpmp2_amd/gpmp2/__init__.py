from .plan_layer import PlanLayer
from .diff_gpmp2_planner import DiffGPMP2Planner
