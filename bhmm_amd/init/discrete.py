"""Initial discrete HMM from raw symbol trajectories.

Host-side numpy; follows bhmm/init/discrete.py:26-338 and bhmm/api.py:231-306: lagged count
matrix -> (reversible) Markov model on the observed symbols -> PCCA+ memberships ->
coarse-grained transition matrix, output probabilities by Bayesian inversion, regularisation.

The reference takes the count matrix, the connectivity helpers and PCCA+ from the unvendored
msmtools package.  They are restated here from their published definitions:
  * sliding-window count matrix, neighbour prior (Prinz et al., J. Chem. Phys. 134, 174105);
  * PCCA+ (Deuflhard & Weber, Lin. Alg. Appl. 398, 161 (2005); Roeblitz & Weber, Adv. Data
    Anal. Classif. 7, 147 (2013)): inner-simplex start, then a Nelder-Mead search over the
    (m-1)^2 free entries of the transformation maximising the crispness trace.
PARITY: everything but PCCA+ is pinned by the golden values of
bhmm/tests/test_init_discrete.py:182-213 (tests/test_host_logic.py); PCCA+ itself is
"parity unpinned" (msmtools is not in the image) and is checked through its defining
properties (memberships are a partition of unity, block structure of metastable chains).
"""
import numpy as np
from scipy.optimize import fmin

from ..estimators import _tmatrix


# ---------------------------------------------------------------- counting
def count_matrix(dtrajs, lag, nstates=None):
    """Sliding-window transition counts at `lag` (msmtools.estimation.count_matrix with its
    defaults, as called at bhmm/api.py:275), dense."""
    if isinstance(dtrajs, np.ndarray) and dtrajs.ndim == 1:
        dtrajs = [dtrajs]
    dtrajs = [np.asarray(d) for d in dtrajs]
    if nstates is None:
        nstates = int(max(int(d.max()) for d in dtrajs if d.size)) + 1
    C = np.zeros((nstates, nstates))
    for d in dtrajs:
        if d.size > lag:
            C += np.bincount(d[:-lag].astype(np.int64) * nstates + d[lag:].astype(np.int64),
                             minlength=nstates * nstates).reshape(nstates, nstates)
    return C


def prior_neighbor(C, alpha=0.001):
    """alpha on every (i, j) that has a count in either direction."""
    C = np.asarray(C, dtype=np.float64)
    return alpha * ((C + C.T) > 0)


def largest_connected_set(C):
    return np.sort(_tmatrix.connected_sets(C, strong=True)[0])


# ---------------------------------------------------------------- PCCA+
def _inner_simplex(evec):
    """Rows of `evec` (n, m) that span the largest simplex (Weber & Galliat 2002): greedy
    Gram-Schmidt farthest-point search.  Returns the m row indices."""
    n, m = evec.shape
    ortho = evec.copy()
    index = np.zeros(m, dtype=int)
    index[0] = int(np.argmax(np.sum(ortho ** 2, axis=1)))
    ortho = ortho - ortho[index[0]][None, :]
    for j in range(1, m):
        if j > 1:
            prev = ortho[index[j - 1]].copy()
            ortho = ortho - np.outer(ortho.dot(prev), prev)
        d = np.sum(ortho ** 2, axis=1)
        index[j] = int(np.argmax(d))
        ortho = ortho / np.sqrt(d[index[j]])
    return index


def _complete_rotation(inner, evec):
    """Fill the first column (rows sum to zero), the first row (memberships >= 0) and the
    overall scale (memberships sum to one) of the transformation around its inner block."""
    m = inner.shape[0] + 1
    rot = np.zeros((m, m))
    rot[1:, 1:] = inner
    rot[1:, 0] = -np.sum(inner, axis=1)
    rot[0, :] = -np.min(evec[:, 1:].dot(rot[1:, :]), axis=0)
    return rot / np.sum(rot[0, :])


def _crispness(vec, evec, m):
    rot = _complete_rotation(vec.reshape(m - 1, m - 1), evec)
    return -np.sum(rot ** 2 / rot[0, :][None, :])


def _pcca_connected(P, m, pi):
    """Memberships (n, m) of an irreducible reversible block."""
    n = P.shape[0]
    if m == 1:
        return np.ones((n, 1))
    if m == n:
        return np.eye(n)
    sq = np.sqrt(pi)
    S = sq[:, None] * P / sq[None, :]
    w, V = np.linalg.eigh(0.5 * (S + S.T))
    order = np.argsort(-w, kind='stable')[:m]
    evec = V[:, order] / sq[:, None]          # right eigenvectors, pi-orthonormal
    evec[:, 0] = 1.0
    for i in range(1, m):
        evec[:, i] /= np.sqrt(np.dot(evec[:, i] * pi, evec[:, i]))
        if evec[np.argmax(np.abs(evec[:, i])), i] < 0:
            evec[:, i] *= -1.0
    index = _inner_simplex(evec)
    rot = np.linalg.inv(evec[index])
    # the simplex search has (m - 1)^2 unknowns and no gradient: beyond ~8 metastable states it
    # costs minutes for a refinement an initial model does not need -- the inner-simplex solution
    # (already a feasible transformation) is used as it is
    if (m - 1) ** 2 <= 64:
        best = fmin(_crispness, rot[1:, 1:].reshape(-1), args=(evec, m), disp=False)
        rot = _complete_rotation(best.reshape(m - 1, m - 1), evec)
    else:
        rot = _complete_rotation(rot[1:, 1:], evec)
    chi = np.clip(evec.dot(rot), 0.0, 1.0)
    return chi / chi.sum(axis=1)[:, None]


def pcca_memberships(P, m):
    """PCCA+ memberships of a reversible transition matrix that may consist of several closed
    sets plus transient states: every closed set gets one metastable state, the remaining
    ones go to the closed sets with the largest sub-dominant eigenvalues, and transient
    states inherit memberships by their absorption probabilities."""
    P = np.asarray(P, dtype=np.float64)
    n = P.shape[0]
    if m > n:
        raise ValueError('Number of metastable states exceeds number of states')
    closed = _tmatrix.closed_sets(P)
    if len(closed) > m:
        raise ValueError('Number of metastable states m = %d is smaller than the number of '
                         'closed sets (%d).' % (m, len(closed)))
    ev = []
    for c, s in enumerate(closed):
        lam = np.sort(np.linalg.eigvals(P[np.ix_(s, s)]).real)[::-1]
        ev += [(lam[k], c) for k in range(1, len(s))]
    ev.sort(key=lambda t: -t[0])
    per_set = np.ones(len(closed), dtype=int)
    for lam, c in ev[:m - len(closed)]:
        per_set[c] += 1
    if per_set.sum() < m:
        raise ValueError('Not enough states in the closed sets for %d metastable states' % m)
    chi = np.zeros((n, m))
    col = 0
    in_closed = np.zeros(n, dtype=bool)
    for c, s in enumerate(closed):
        Pss = P[np.ix_(s, s)]
        chi[s, col:col + per_set[c]] = _pcca_connected(Pss, per_set[c],
                                                       _tmatrix.stationary_vector(Pss))
        col += per_set[c]
        in_closed[s] = True
    t = np.where(~in_closed)[0]
    if t.size:
        c = np.where(in_closed)[0]
        chi[t] = np.linalg.solve(np.eye(t.size) - P[np.ix_(t, t)], P[np.ix_(t, c)].dot(chi[c]))
        chi[t] = np.clip(chi[t], 0.0, None)
        chi[t] /= chi[t].sum(axis=1)[:, None]
    return chi


# ---------------------------------------------------------------- coarse graining
def coarse_grain_transition_matrix(P, M):
    """bhmm/init/discrete.py:26-56: (M'M)^-1 M' P M, negatives clipped, rows normalised."""
    W = np.linalg.inv(np.dot(M.T, M))
    Pc = np.maximum(np.dot(W, np.dot(np.dot(M.T, P), M)), 0)
    return Pc / Pc.sum(axis=1)[:, None]


def regularize_hidden(p0, P, reversible=True, stationary=False, C=None, eps=None):
    """bhmm/init/discrete.py:59-115 (with stationary=True the initial distribution is left
    as given there, :108-109)."""
    n = P.shape[0]
    if eps is None:
        eps = 0.01 / n
    P = np.maximum(P, eps)
    P = P / P.sum(axis=1)[:, None]
    if reversible:
        P = _tmatrix.enforce_reversible_on_closed(P)
    if not stationary:
        p0 = np.maximum(p0, eps)
        p0 = p0 / p0.sum()
    return p0, P


def regularize_pobs(B, nonempty=None, separate=None, eps=None):
    """bhmm/init/discrete.py:118-165."""
    B = np.array(B, dtype=np.float64)
    n, m = B.shape
    if eps is None:
        eps = 0.01 / m
    if nonempty is None:
        nonempty = np.arange(m)
    if separate is None:
        B[:, nonempty] = np.maximum(B[:, nonempty], eps)
    else:
        ns = np.array(sorted(set(nonempty) - set(separate)), dtype=int)
        sp = np.array(sorted(set(nonempty) & set(separate)), dtype=int)
        B[:n - 1, ns] = np.maximum(B[:n - 1, ns], eps)
        B[n - 1, sp] = np.maximum(B[n - 1, sp], eps)
    return B / B.sum(axis=1)[:, None]


def init_discrete_hmm_spectral(C_full, nstates, reversible=True, stationary=True,
                               active_set=None, P=None, eps_A=None, eps_B=None, separate=None):
    """bhmm/init/discrete.py:168-338.  Returns (p0, A, B)."""
    C_full = np.asarray(C_full, dtype=np.float64)
    nfull = C_full.shape[0]
    if eps_A is None:
        eps_A = 0.01 / nstates
    if eps_B is None:
        eps_B = 0.01 / nfull
    symsum = C_full.sum(axis=0) + C_full.sum(axis=1)
    nonempty = np.where(symsum > 0)[0]
    if active_set is None:
        active_set = nonempty
    else:
        active_set = np.asarray(active_set, dtype=int)
        if np.any(symsum[active_set] == 0):
            raise ValueError('Given active set has empty states')
    if P is not None and np.shape(P)[0] != active_set.size:
        raise ValueError('Given initial transition matrix P has shape ' + str(np.shape(P))
                         + 'while active set has size ' + str(active_set.size))
    if separate is None:
        active_nonseparate = active_set.copy()
        nmeta = nstates
    else:
        if np.max(separate) >= nfull:
            raise ValueError('Separate set has indexes that do not exist in full state space: '
                             + str(np.max(separate)))
        active_nonseparate = np.array(sorted(set(active_set) - set(separate)), dtype=int)
        nmeta = nstates - 1
    if active_nonseparate.size < nmeta:
        raise NotImplementedError('Trying to initialize ' + str(nmeta) + '-state HMM from smaller '
                                  + str(active_nonseparate.size) + '-state MSM.')

    C_active = C_full[np.ix_(active_set, active_set)]
    P_active = _tmatrix.estimate_P(C_active, reversible=reversible, maxiter=10000) \
        if P is None else np.asarray(P, dtype=np.float64)
    pi_active = _tmatrix.stationary_distribution(P_active, C=C_active)
    pi_full = np.zeros(nfull)
    pi_full[active_set] = pi_active

    C_ans = C_full[np.ix_(active_nonseparate, active_nonseparate)]
    if reversible and separate is None:
        P_ans = P_active
    else:
        P_ans = _tmatrix.estimate_P(C_ans, reversible=True)

    if active_nonseparate.size > nmeta:
        # the reference's PCCA object starts from msmtools' stationary distribution, which refuses
        # a matrix that is not (weakly) connected: estimation on disconnected data raises
        # ValueError there (bhmm/tests/test_mlhmm.py:127-140), and so does this
        if not _tmatrix.is_connected(P_ans, strong=False):
            raise ValueError('Input matrix is not weakly connected. Therefore it has no unique '
                             'stationary distribution.')
        M_ans = pcca_memberships(P_ans, nmeta)
        w = M_ans * _tmatrix.stationary_distribution(P_ans, C=C_ans)[:, None]
        B_ans = (w / w.sum(axis=0)[None, :]).T
    else:
        M_ans = np.eye(nmeta)
        B_ans = np.eye(nmeta)

    if separate is None:
        M_active = M_ans
    else:
        M_full = np.zeros((nfull, nstates))
        M_full[active_nonseparate, :nmeta] = M_ans
        M_full[np.asarray(separate, dtype=int), -1] = 1
        M_active = M_full[active_set]

    P_hmm = coarse_grain_transition_matrix(P_active, M_active)
    if reversible:
        P_hmm = _tmatrix.enforce_reversible_on_closed(P_hmm)
    C_hmm = M_active.T.dot(C_active).dot(M_active)
    pi_hmm = _tmatrix.stationary_distribution(P_hmm, C=C_hmm)

    B_hmm = np.zeros((nstates, nfull))
    B_hmm[:nmeta, active_nonseparate] = B_ans
    if separate is not None:
        B_hmm[-1, np.asarray(separate, dtype=int)] = pi_full[np.asarray(separate, dtype=int)]

    pi_hmm, P_hmm = regularize_hidden(pi_hmm, P_hmm, reversible=reversible,
                                      stationary=stationary, C=C_hmm, eps=eps_A)
    B_hmm = regularize_pobs(B_hmm, nonempty=nonempty, separate=separate, eps=eps_B)
    return pi_hmm, P_hmm, B_hmm


def init_discrete_hmm(observations, nstates, lag=1, reversible=True, stationary=True,
                      regularize=True, method='connect-spectral', separate=None):
    """bhmm/api.py:231-306.  Returns (p0, A, B); bhmm_amd.api wraps them in an HMM."""
    C = count_matrix(observations, lag)
    eps_A = eps_B = None if regularize else 0
    if not stationary:
        raise NotImplementedError('Discrete-HMM initialization with stationary=False is not yet '
                                  'implemented.')
    if method == 'lcs-spectral':
        active = largest_connected_set(C)
    elif method == 'connect-spectral':
        C = C + prior_neighbor(C, 0.001)
        active = np.where(C.sum(axis=0) + C.sum(axis=1) > 0)[0]
        C[active, active] = np.maximum(C[active, active], 0.001)
    elif method == 'spectral':
        active = None
    else:
        raise NotImplementedError('Unknown discrete-HMM initialization method ' + str(method))
    return init_discrete_hmm_spectral(C, nstates, reversible=reversible, stationary=stationary,
                                      active_set=active, separate=separate, eps_A=eps_A,
                                      eps_B=eps_B)
