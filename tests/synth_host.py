"""Host restatement of bhmm_synth_observations (bhmm_amd/csrc/synth_api.hip) -- TEST
INFRASTRUCTURE: the same counter-based stream and inverse-CDF draws in numpy, so that the
device-generated benchmark data can be checked (discrete: bit for bit) and small samples of a big
device-generated workload can be reproduced on the host without copying them back."""
import numpy as np

from oracle_engine import device_uniforms


def _pick(cdf, u):
    """first j with u < cdf[j], last index if none (cdf_pick)."""
    n = len(cdf)
    return min(int(np.searchsorted(cdf[:n - 1], u, side='right')), n - 1)


def synth_discrete(A, pi, B, k, T, seed, K_T=None):
    """Trajectory k of a (K, T) batch: returns (obs int32 (T,), states uint8 (T,))."""
    n, M = B.shape
    cA = np.cumsum(A, axis=1)
    cpi = np.cumsum(pi)
    cB = np.cumsum(B, axis=1)
    u = device_uniforms(seed, 2 * k * T, 2 * T)
    us, ue = u[0::2], u[1::2]
    s = np.empty(T, dtype=np.uint8)
    o = np.empty(T, dtype=np.int32)
    cur = 0
    for t in range(T):
        cur = _pick(cpi, us[t]) if t == 0 else _pick(cA[cur], us[t])
        s[t] = cur
        o[t] = _pick(cB[cur], ue[t])
    return o, s


def synth_gaussian(A, pi, mu, sigma, k, T, seed):
    n = len(mu)
    cA = np.cumsum(A, axis=1)
    cpi = np.cumsum(pi)
    u = device_uniforms(seed, 2 * k * T, 2 * T)
    us, ue = u[0::2], u[1::2]
    u2 = device_uniforms(seed ^ 0xD1B54A32D192ED03, 2 * k * T, 2 * T)[1::2]
    s = np.empty(T, dtype=np.uint8)
    cur = 0
    for t in range(T):
        cur = _pick(cpi, us[t]) if t == 0 else _pick(cA[cur], us[t])
        s[t] = cur
    r = np.sqrt(-2.0 * np.log(1.0 - ue))
    o = np.asarray(mu)[s] + np.asarray(sigma)[s] * r * np.cos(6.283185307179586 * u2)
    return o, s
