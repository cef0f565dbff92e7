"""qpwcnet_amd -- MI355X (gfx950) hot path of yycho0108/qpwcnet.

CostVolume + bilinear Warp as hand-written HIP kernels behind the reference's
layer surface; the surrounding pyramid / flow-estimator convolutions run on
PyTorch-ROCm.  See DESIGN.md.
"""
from .backend import image_data_format, set_image_data_format  # noqa: F401

__version__ = "0.1.0"
