"""Import-time stand-in for OpenCV (absent in this image).

TEST INFRASTRUCTURE ONLY.  The reference (read-only, /root/reference) imports cv2
at module scope for constant tables (codes/options/options.py:4,11-36,
codes/dataops/common.py:7, codes/utils/util.py:5).  No cv2 *function* is on the
SR training hot path, so upper-case names resolve to distinct ints and anything
else raises.
"""
_consts = {}
__version__ = "4.5.5"     # extra_functional.py:23 compares the version at import time


def __getattr__(name):
    if name.isupper() or (name[:1].isupper() and "_" in name):
        if name not in _consts:
            _consts[name] = 1000 + len(_consts)
        return _consts[name]
    raise AttributeError("cv2 stub: %s is not available" % name)


# ---- the two OpenCV primitives of the reference's SSIM (codes/utils/metrics.py:180-199), on independent installed implementations, so
# that oracle/make_golden_metrics.py can run the reference's OWN ssim / calculate_ssim code:
def getGaussianKernel(ksize, sigma, ktype=None):
    """cv2.getGaussianKernel for sigma > 0: exp(-(i - (ksize - 1) / 2)^2 / (2 sigma^2)), normalised to sum 1, as a [ksize, 1] column
    (OpenCV's fixed small-kernel tables only apply to sigma <= 0)."""
    import numpy as np
    from scipy.signal.windows import gaussian
    if sigma <= 0:
        raise NotImplementedError("cv2 stub: getGaussianKernel needs sigma > 0")
    k = gaussian(int(ksize), float(sigma)).astype(np.float64)
    return (k / k.sum()).reshape(-1, 1)


def filter2D(src, ddepth, kernel):
    """cv2.filter2D(src, -1, kernel): correlation with a centred anchor and BORDER_REFLECT_101 (scipy.ndimage 'mirror'), per channel."""
    import numpy as np
    from scipy import ndimage
    if ddepth != -1:
        raise NotImplementedError("cv2 stub: filter2D only with ddepth = -1")
    src = np.asarray(src)
    k = np.asarray(kernel, dtype=np.float64)
    if src.ndim == 2:
        return ndimage.correlate(src, k, mode="mirror")
    return np.stack([ndimage.correlate(src[..., c], k, mode="mirror") for c in range(src.shape[2])], axis=-1)
