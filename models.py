"""Drop-in for the reference's top-level `models` module (inference path only).

The reference drivers do `from models import CVM_VIGOR as CVM` / `CVM_VIGOR_ori_prior` / `CVM_KITTI` /
`CVM_OxfordRobotCar` (train_VIGOR.py:17-18, train_KITTI.py:17, train_OxfordRobotCar.py:17); with this
repository's root on sys.path those imports resolve to the MI355X implementation.
"""
from ccvpe_amd.models import CVM_KITTI, CVM_OxfordRobotCar, CVM_VIGOR, CVM_VIGOR_ori_prior  # noqa: F401

__all__ = ["CVM_VIGOR", "CVM_VIGOR_ori_prior", "CVM_KITTI", "CVM_OxfordRobotCar"]
