"""Mask morphology and the Canny edge filter of source_setup, behind the reference's names
(iPERCore/tools/utils/morphology/{morph_ops,canny_ops}.py), executed by csrc/source.hip.

``morph(mask, ks, mode)`` / ``soft_dilate(mask, ks)`` keep the reference signatures (the ``kernel=`` argument of
the reference is always the all-ones default on this path).  ``CannyFilter()(img, low, high, hysteresis=True)``
returns the thin-edge map only (``make_morph_image`` - its single caller, flowcomposition.py:350-352 - discards
the other five outputs).
"""
import numpy as np

from . import ops


def morph(src_bg_mask, ks, mode="erode", kernel=None):
    if kernel is not None:
        raise NotImplementedError("only the all-ones structuring element of the reference's call sites is built")
    return ops.morph(src_bg_mask, ks, "erode" if mode == "erode" else "dilate")


def soft_dilate(src_bg_mask, ks, kernel=None):
    if kernel is not None:
        raise NotImplementedError("only the all-ones structuring element is built")
    return ops.morph(src_bg_mask, ks, "soft_dilate")


def gaussian_kernel(k=3, mu=0, sigma=1):
    """canny_ops.py:9-24 (float64, as the reference builds it before the float32 cast)."""
    g = np.linspace(-1, 1, k)
    x, y = np.meshgrid(g, g)
    d = (x ** 2 + y ** 2) ** 0.5
    k2 = np.exp(-(d - mu) ** 2 / (2 * sigma ** 2)) / (2 * np.pi * sigma ** 2)
    return k2 / np.sum(k2)


def sobel_kernel(k=3):
    """canny_ops.py:27-36."""
    r = np.linspace(-(k // 2), k // 2, k)
    x, y = np.meshgrid(r, r)
    den = x ** 2 + y ** 2
    den[:, k // 2] = 1
    return x / den


class CannyFilter(object):
    def __init__(self, k_gaussian=3, mu=0, sigma=1, k_sobel=3, device=None):
        if k_gaussian != 3 or k_sobel != 3:
            raise NotImplementedError("3x3 gaussian / sobel (the reference defaults) only")
        self.gauss9 = gaussian_kernel(k_gaussian, mu, sigma).astype(np.float32).reshape(-1).tolist()
        self.sobelx9 = sobel_kernel(k_sobel).astype(np.float32).reshape(-1).tolist()

    def to(self, device):
        return self

    def __call__(self, img, low_threshold=None, high_threshold=None, hysteresis=False):
        if img.shape[1] != 1 or low_threshold is None or high_threshold is None or not hysteresis:
            raise NotImplementedError("built for the reference's call: 1-channel silhouette, (low, high, hysteresis=True)")
        return ops.canny_edges(img, self.gauss9, self.sobelx9, low_threshold, high_threshold)
