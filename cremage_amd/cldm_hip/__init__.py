"""ControlNet on the HIP kernels: drop-ins for the reference's `cldm.cldm.{ControlNet, ControlledUnetModel, ControlLDM}`
(modules/cldm/cldm.py), reachable through the same YAML `target:` lines (configs/cldm_v15-hip.yaml)."""
from .cldm import ControlLDM, ControlledUnetModel, ControlNet  # noqa: F401
