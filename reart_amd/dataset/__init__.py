"""Data formats on the input side of the path (the reference's ``dataset/dataset_robot.py``)."""
from .dataset_robot import Sequence  # noqa: F401
