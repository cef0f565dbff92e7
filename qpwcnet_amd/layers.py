"""``torch.nn.Module`` twins of the reference's Keras hot-path layers
(qpwcnet/core/layers.py:32-186): ``CostVolume``, ``CostVolumeV2``, ``Warp``,
``WarpV2``.  Same names, same constructor arguments, same ``layer((a, b))``
call convention, same config round trip; the arithmetic runs in the gfx950 HIP
kernels behind ``include/qpwc.h``.

Layout is read from the process-global ``image_data_format()`` at construction
time exactly like the reference (layers.py:41,119,146,173); an explicit
``data_format=`` keyword (what the reference's test scripts try to pass,
test/test_cost_volume.py:10-11, test/test_warp.py:14-15) overrides it.
"""
import torch

from . import ops
from .backend import CHANNELS_FIRST, CHANNELS_LAST, get_axis, image_data_format


def lrelu(x):
    """qpwcnet/core/layers.py:15-16."""
    return torch.nn.functional.leaky_relu(x, 0.1)


def _get_axis(data_format):
    return get_axis(data_format)


class _HotPathLayer(torch.nn.Module):
    def __init__(self, *args, data_format=None, name=None, **kwargs):
        if args or kwargs:
            # Keras' Layer.__init__ rejects unknown arguments as well
            raise TypeError("unexpected arguments: {} {}".format(args, sorted(kwargs)))
        super().__init__()
        self.data_format = image_data_format() if data_format is None else data_format
        self.axis = _get_axis(self.data_format)  # ValueError('Unsupported data format : ...')
        self.layer_name = name
        self.h = None
        self.w = None

    def build(self, input_shapes):
        """Captures H, W from the first input's shape (layers.py:57-70, 153-164)."""
        shape = input_shapes[0]
        if self.data_format == CHANNELS_FIRST:
            self.h, self.w = shape[2], shape[3]
        elif self.data_format == CHANNELS_LAST:
            self.h, self.w = shape[1], shape[2]
        else:
            raise ValueError("Unsupported data format : {}".format(self.data_format))

    def _unpack(self, inputs):
        a, b = inputs
        self.build((tuple(a.shape), tuple(b.shape)))
        return a, b

    def call(self, inputs):  # Keras spelling
        return self.forward(inputs)

    def get_config(self):
        cfg = {"name": self.layer_name}
        cfg.update(getattr(self, "_config", {}))
        return cfg

    @classmethod
    def from_config(cls, config):
        return cls(**config)


class CostVolume(_HotPathLayer):
    """qpwcnet/core/layers.py:32-109 -- pure-TF cost volume; here the HIP kernel."""

    def __init__(self, search_range=4, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._config = {"search_range": search_range}
        self.search_range = search_range

    def forward(self, inputs):
        prv, nxt = self._unpack(inputs)
        return ops.cost_volume(prv, nxt, self.search_range, self.data_format, 0.1)


class CostVolumeV2(CostVolume):
    """qpwcnet/core/layers.py:112-141 -- tfa CorrelationCost(1, r, 1, 1, r) + lrelu.
    Identical function to CostVolume (app/test/test_cvol_equal.py:25): same kernel."""


class Warp(_HotPathLayer):
    """qpwcnet/core/layers.py:144-168 -> tf_warp (qpwcnet/core/warp.py:63-153)."""

    def forward(self, inputs):
        img, flo = self._unpack(inputs)
        return ops.warp(img, flo, "tfwarp", self.data_format)


class WarpV2(_HotPathLayer):
    """qpwcnet/core/layers.py:171-186 -> tfa.image.dense_image_warp(img, -flo[..., ::-1])."""

    def forward(self, inputs):
        img, flo = self._unpack(inputs)
        return ops.warp(img, flo, "clamp", self.data_format)
