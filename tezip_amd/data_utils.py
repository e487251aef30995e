"""Padding helpers with the reference's names and results (/root/reference/src/data_utils.py:77-107).
Only shapes are computed on the host: the zero padding itself happens inside the HIP kernels
(key frames are read unpadded and treated as 0 outside H x W)."""


def padding_size(num):
    """data_utils.py:103-107."""
    if num % 8 == 0:
        return num
    return (int(num / 8) + 1) * 8


def padding_shape(height, width):
    """data_utils.py:94-100."""
    return (padding_size(height), padding_size(width))
