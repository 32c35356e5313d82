"""localrf_amd -- MI355X-native render path of facebookresearch/localrf.

Public surface mirrors the reference's names:
  TensorVMSplit, AlphaGridMask      (models/tensoRF.py, models/tensorBase.py)
  LocalTensorfs                     (local_tensorfs.py)
  rays.*                            (utils/ray_utils.py, utils/utils.py 6D helpers)
  FusedAdam                         (torch.optim.Adam as local_tensorfs.py:88-97,146 configures it)
  losses.flow_loss / depth_loss     (train.py:385-423 with utils/utils.py:15-59)
The arithmetic of TensorVMSplit.forward and of LocalTensorfs.forward (ray generation, field
blend, exposure) runs in hand-written HIP kernels for gfx950 (csrc/), reached through the C ABI
of include/lrf.h.
"""
from ._native import NativeError  # noqa: F401
from .field import TensorVMSplit, AlphaGridMask, MLPRender_Fea_late_view  # noqa: F401
from .scene import LocalTensorfs  # noqa: F401
from .optim import FusedAdam  # noqa: F401
from . import rays  # noqa: F401
from . import losses  # noqa: F401

__all__ = ["TensorVMSplit", "AlphaGridMask", "MLPRender_Fea_late_view", "LocalTensorfs", "rays", "losses", "NativeError", "FusedAdam"]
