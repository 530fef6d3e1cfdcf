"""Sample statistics used by SampledHMM: bhmm/util/statistics.py:34-151."""
import math

import numpy as np


def confidence_interval(data, alpha):
    """Mean and the interval that holds the fraction `alpha` of the sorted samples around it,
    with linear interpolation between order statistics (statistics.py:34-72)."""
    if alpha < 0 or alpha > 1:
        raise ValueError('Not a meaningful confidence level: ' + str(alpha))
    data = np.asarray(data, dtype=np.float64)
    m = np.mean(data)
    s = np.sort(data)
    n = len(s)
    im = int(np.searchsorted(s, m))
    if im == 0 or im == n:
        pm = im
    else:
        pm = (im - 1) + (m - s[im - 1]) / (s[im] - s[im - 1])

    def at(p):
        i1 = max(0, int(math.floor(p)))
        i2 = min(n - 1, int(math.ceil(p)))
        return s[i1] + (p - i1) * (s[i2] - s[i1])

    return m, at(pm - alpha * pm), at(pm + alpha * (n - im))


def confidence_interval_arr(data, conf=0.95):
    """Element-wise (lower, upper) over the leading sample axis (statistics.py:103-151)."""
    if conf < 0 or conf > 1:
        raise ValueError('Not a meaningful confidence level: ' + str(conf))
    data = np.array([np.asarray(d, dtype=np.float64) for d in data])
    if data.ndim < 2 or data.ndim > 3:
        raise NotImplementedError('Only supporting arrays of dimension 1 and 2 as yet.')
    lower = np.zeros(data.shape[1:])
    upper = np.zeros(data.shape[1:])
    for idx in np.ndindex(*data.shape[1:]):
        _, lower[idx], upper[idx] = confidence_interval(data[(slice(None),) + idx], conf)
    return lower, upper
