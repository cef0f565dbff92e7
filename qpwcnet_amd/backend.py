"""Process-global image data format, mirroring ``tf.keras.backend.image_data_format``
which the reference layers read at construction time
(qpwcnet/core/layers.py:41,119,146,173; qpwcnet/core/non_layers.py:60,113,128,141)."""

CHANNELS_LAST = "channels_last"
CHANNELS_FIRST = "channels_first"

_IMAGE_DATA_FORMAT = CHANNELS_LAST  # Keras default


def image_data_format():
    return _IMAGE_DATA_FORMAT


def set_image_data_format(data_format):
    global _IMAGE_DATA_FORMAT
    if data_format not in (CHANNELS_LAST, CHANNELS_FIRST):
        raise ValueError("Unknown data_format: {}".format(data_format))
    _IMAGE_DATA_FORMAT = data_format


def get_axis(data_format):
    """Channel axis of a rank-4 tensor -- ``_get_axis``, qpwcnet/core/layers.py:19-29."""
    if data_format == CHANNELS_FIRST:
        return 1
    if data_format == CHANNELS_LAST:
        return 3
    raise ValueError("Unsupported data format : {}".format(data_format))
