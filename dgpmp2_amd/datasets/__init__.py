from .planning_dataset import PlanningDataset, write_environment, write_problem, write_meta
