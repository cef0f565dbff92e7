"""The one function of the reference's ``qpwcnet/core/vis.py`` that touches the hot path's data:
``cost_volume_to_flow`` (vis.py:9-34) decodes a cost volume into the displacement of its strongest
correlation -- the in-tree statement of the channel order ``i0 * 9 + j0``.  (``flow_to_image`` and the
other drawing helpers are visualisation and out of scope)"""
from . import ops
from .backend import image_data_format


def cost_volume_to_flow(cvol, data_format=None):
    if data_format is None:
        data_format = image_data_format()
    return ops.cost_volume_to_flow(cvol, data_format)
