"""Host-side transition-matrix estimation for the M-step: the O(N^2..N^3) work of
bhmm/estimators/_tmatrix_disconnected.py, whose arithmetic the reference delegates to the
(unvendored) msmtools package.

What is pinned and what is not: the non-reversible branch is exactly
row-normalisation with empty rows set to C_ii = 1 (_tmatrix_disconnected.py:107-115).  The
reversible maximum-likelihood estimator is restated from the published fixed-point iteration
(Bowman et al. 2009; Prinz et al. 2011, Eq. 29-31; Trendelkamp-Schroer et al. 2015, Alg. 1)
and is "parity unpinned" against msmtools itself; it is checked by self-consistency
(detailed balance, likelihood optimality) and the closed forms of
bhmm/tests/test_mlhmm_patho.py.  The partially reversible iteration restates
_tmatrix_disconnected.py:126-190.
"""
import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import connected_components


def is_transition_matrix(T, tol=1e-10):
    T = np.asarray(T)
    if T.ndim != 2 or T.shape[0] != T.shape[1]:
        return False
    # (numpy.allclose(rowsums, 1, atol=1e-8) is |rowsum - 1| <= 1e-8 + 1e-5, NaN fails; written out
    # because the generic routine costs 40 us and this sits in every EM iteration / Gibbs sweep)
    if not T.min() >= -tol:
        return False
    dev = np.abs(T.sum(axis=1) - 1.0).max()
    return bool(dev <= 1e-8 + 1e-5)


def _component_labels_dense(adj, strong):
    """Component label (smallest member) of every node of a small dense graph: reachability by
    repeated squaring of the boolean adjacency matrix.  The per-EM-iteration M-step calls this
    several times on an N x N matrix; scipy's sparse-graph routine costs 0.2 ms per call there."""
    n = adj.shape[0]
    if not strong:
        adj = adj | adj.T
    reach = (adj | np.eye(n, dtype=bool)).astype(np.float32)
    while True:
        nxt = ((reach @ reach) > 0).astype(np.float32)
        if np.array_equal(nxt, reach):
            break
        reach = nxt
    mutual = (reach > 0) & (reach.T > 0)
    return mutual.argmax(axis=1)


def connected_sets(C, mincount_connectivity=0, strong=True):
    """_tmatrix_disconnected.py:28-43: sets sorted by decreasing size."""
    Cc = np.array(C, dtype=np.float64)
    Cc[Cc <= mincount_connectivity] = 0
    if Cc.shape[0] <= 128:
        labels = _component_labels_dense(Cc > 0, strong)
        sets = [np.where(labels == i)[0] for i in np.unique(labels)]
    else:
        n, labels = connected_components(csr_matrix(Cc), directed=True,
                                         connection='strong' if strong else 'weak')
        sets = [np.where(labels == i)[0] for i in range(n)]
    sets.sort(key=lambda s: (-len(s), s[0]))
    return sets


def is_connected(C, mincount_connectivity=0, strong=True):
    return len(connected_sets(C, mincount_connectivity=mincount_connectivity, strong=strong)) == 1


def closed_sets(C, mincount_connectivity=0):
    """_tmatrix_disconnected.py:46-56: strongly connected sets without outgoing counts."""
    C = np.asarray(C, dtype=np.float64)
    n = C.shape[0]
    closed = []
    for s in connected_sets(C, mincount_connectivity=mincount_connectivity, strong=True):
        mask = np.zeros(n, dtype=bool)
        mask[s] = True
        if C[np.ix_(mask, ~mask)].sum() == 0:
            closed.append(s)
    return closed


def nonempty_set(C, mincount_connectivity=0):
    """_tmatrix_disconnected.py:59-65."""
    C = np.asarray(C, dtype=np.float64)
    if mincount_connectivity > 0:
        C = np.where(C < mincount_connectivity, 0.0, C)
    return np.where(C.sum(axis=0) + C.sum(axis=1) > 0)[0]


def stationary_vector(P):
    """Stationary distribution of an irreducible stochastic matrix (dense eigen-solve with a
    linear-system fallback)."""
    P = np.asarray(P, dtype=np.float64)
    n = P.shape[0]
    if n == 1:
        return np.ones(1)
    A = np.vstack([P.T - np.eye(n), np.ones((1, n))])
    b = np.zeros(n + 1)
    b[-1] = 1.0
    pi = np.linalg.lstsq(A, b, rcond=None)[0]
    pi = np.maximum(pi, 0.0)
    return pi / pi.sum()


NATIVE = True      # mle_reversible: run the fixed point in the library (tests switch it off to
                   # compare the library with the pure-numpy loop)


def mle_reversible(C, maxiter=1000000, maxerr=1e-8, native=None):
    """Reversible maximum-likelihood transition matrix of a strongly connected count matrix.
    Fixed point  x_ij <- (c_ij + c_ji) / (c_i / x_i + c_j / x_j),  P_ij = x_ij / x_i.
    native=True runs the same iteration in the library (bhmm_mle_reversible, host code: thousands
    of O(n^2) iterations cost tens of milliseconds in numpy); the numpy loop below is the
    restatement it is tested against, and what runs if the library is not built."""
    C = np.asarray(C, dtype=np.float64)
    if native is None:
        native = NATIVE
    if native and C.ndim == 2 and C.shape[0] == C.shape[1] and np.all(np.isfinite(C)) \
            and C.sum() > 0:
        try:
            from .. import _lib
            L = _lib.load()
        except (ImportError, OSError, AttributeError):
            L = None
        if L is not None:
            Cc = np.ascontiguousarray(C)
            P = np.empty_like(Cc)
            _lib.check(L.bhmm_mle_reversible(_lib.dp(P), None, _lib.dp(Cc), int(Cc.shape[0]),
                                             int(maxiter), float(maxerr)))
            return P
    C2 = C + C.T
    csum = C.sum(axis=1)
    X = C2 / C2.sum()
    xsum = X.sum(axis=1)
    it, err = 0, 1.0
    while err > maxerr and it < maxiter:
        q = csum / xsum
        with np.errstate(divide='ignore', invalid='ignore'):
            X = C2 / (q[:, None] + q[None, :])
        X[C2 == 0] = 0.0
        X /= X.sum()
        xnew = X.sum(axis=1)
        err = np.max(np.abs(xnew - xsum))
        xsum = xnew
        it += 1
    return X / X.sum(axis=1)[:, None]


def mle_reversible_fixed_pi(C, pi, maxiter=1000000, maxerr=1e-8):
    """Reversible MLE with a given stationary vector (Trendelkamp-Schroer & Noe 2013):
    Lagrange-multiplier fixed point."""
    C = np.asarray(C, dtype=np.float64)
    pi = np.asarray(pi, dtype=np.float64)
    n = C.shape[0]
    C2 = C + C.T
    lam = 0.5 * C2.sum(axis=1)
    lam[lam == 0] = 1.0
    it, err = 0, 1.0
    while err > maxerr and it < maxiter:
        D = lam[:, None] * pi[None, :] + lam[None, :] * pi[:, None]
        with np.errstate(divide='ignore', invalid='ignore'):
            F = np.where(C2 > 0, C2 * pi[None, :] * lam[:, None] / D, 0.0)
        lam_new = F.sum(axis=1)
        lam_new[lam_new == 0] = lam[lam_new == 0]
        err = np.max(np.abs(lam_new - lam) / np.maximum(lam, 1e-300))
        lam = lam_new
        it += 1
    D = lam[:, None] * pi[None, :] + lam[None, :] * pi[:, None]
    with np.errstate(divide='ignore', invalid='ignore'):
        P = np.where(C2 > 0, C2 * pi[None, :] / D, 0.0)
    P[np.arange(n), np.arange(n)] = 0.0
    P[np.arange(n), np.arange(n)] = 1.0 - P.sum(axis=1)
    return P


def transition_matrix_partial_rev(C, P, S, maxiter=1000000, maxerr=1e-8):
    """_tmatrix_disconnected.py:126-190: rows S reversible among themselves, with outgoing
    counts to the rest.  Writes rows S of P."""
    S = np.asarray(S, dtype=bool)
    Css = C[S][:, S]
    Cso = C[S][:, ~S]
    sym = Css + Css.T
    rowcounts = C[S].sum(axis=1)
    X = 0.5 * sym
    Y = Cso.copy()
    tot = X.sum() + Y.sum()
    X, Y = X / tot, Y / tot
    rows = X.sum(axis=1) + Y.sum(axis=1)
    it, err = 0, 1.0
    while err > maxerr and it < maxiter:
        d = rowcounts / rows
        X = sym / (d[:, None] + d)
        Y = Cso / d[:, None]
        tot = X.sum() + Y.sum()
        X, Y = X / tot, Y / tot
        new_rows = X.sum(axis=1) + Y.sum(axis=1)
        err = np.max(np.abs(new_rows - rows))
        rows = new_rows
        it += 1
    P[np.ix_(S, S)] = X
    P[np.ix_(S, ~S)] = Y
    P[S] /= P[S].sum(axis=1)[:, None]


def estimate_P(C, reversible=True, fixed_statdist=None, maxiter=1000000, maxerr=1e-8,
               mincount_connectivity=0):
    """_tmatrix_disconnected.py:68-123."""
    C = np.asarray(C, dtype=np.float64)
    n = C.shape[0]
    P = np.eye(n)
    if reversible and fixed_statdist is None:
        for s in connected_sets(C, mincount_connectivity=mincount_connectivity, strong=True):
            mask = np.zeros(n, dtype=bool)
            mask[s] = True
            if C[np.ix_(mask, ~mask)].sum() > np.finfo(C.dtype).eps:
                transition_matrix_partial_rev(C, P, mask, maxiter=maxiter, maxerr=maxerr)
            elif s.size > 1:
                I = np.ix_(mask, mask)
                P[I] = mle_reversible(C[I], maxiter=maxiter, maxerr=maxerr)
    else:
        for s in connected_sets(C, mincount_connectivity=mincount_connectivity, strong=False):
            I = np.ix_(s, s)
            if not reversible:
                Csub = C[I].copy()
                zero_rows = np.where(Csub.sum(axis=1) == 0)[0]
                Csub[zero_rows, zero_rows] = 1.0
                P[I] = Csub / Csub.sum(axis=1)[:, None]
            elif fixed_statdist is not None:
                pi_s = np.asarray(fixed_statdist)[s]
                P[I] = mle_reversible_fixed_pi(C[I], pi_s / pi_s.sum(), maxiter=maxiter,
                                               maxerr=maxerr)
            else:
                raise NotImplementedError('Transition estimation for the case reversible='
                                          + str(reversible) + ' not implemented.')
    return P


def stationary_distribution(P, C=None, mincount_connectivity=0):
    """_tmatrix_disconnected.py:229-251."""
    if C is None:
        if is_connected(P, strong=True):
            return stationary_vector(P)
        raise ValueError('Computing stationary distribution for disconnected matrix. '
                         'Need count matrix.')
    C = np.asarray(C, dtype=np.float64)
    pi = np.zeros(C.shape[0])
    ctot = C.sum()
    for s in connected_sets(C, mincount_connectivity=mincount_connectivity, strong=False):
        w = C[s, :].sum() / ctot
        pi[s] = w * stationary_vector(P[s, :][:, s])
    return pi / pi.sum()


def enforce_reversible_on_closed(P):
    """_tmatrix_disconnected.py:193-209: symmetrise the stationary flux of every closed set."""
    P = np.asarray(P, dtype=np.float64)
    Prev = P.copy()
    for s in closed_sets(P):
        I = np.ix_(s, s)
        X = stationary_vector(P[I])[:, None] * P[I]
        X = 0.5 * (X + X.T)
        Prev[I] = X / X.sum(axis=1)[:, None]
    return Prev


def _rdl_block(P, reversible):
    """Right eigenvectors (columns), eigenvalues, left eigenvectors (rows) of an irreducible
    block, sorted by decreasing modulus; L R = 1, L[0] = stationary vector, R[:, 0] = 1
    (the 'reversible' / 'standard' normalisations of msmtools.analysis.rdl_decomposition)."""
    if reversible:
        pi = stationary_vector(P)
        sq = np.sqrt(pi)
        S = sq[:, None] * P / sq[None, :]
        w, V = np.linalg.eigh(0.5 * (S + S.T))
        order = np.argsort(-np.abs(w), kind='stable')
        w, V = w[order], V[:, order]
        R = V / sq[:, None]
        L = (V * sq[:, None]).T
        s0 = L[0].sum()
        L[0] /= s0
        R[:, 0] *= s0
        for i in range(1, len(w)):       # fix the sign: largest component of R positive
            if R[np.argmax(np.abs(R[:, i])), i] < 0:
                R[:, i] *= -1.0
                L[i] *= -1.0
        return R, w, L
    w, R = np.linalg.eig(P)
    order = np.argsort(-np.abs(w), kind='stable')
    w, R = w[order], R[:, order]
    L = np.linalg.inv(R)
    s0 = L[0].sum()
    L[0] /= s0
    R[:, 0] *= s0
    return R, w, L


def rdl_decomposition(P, reversible=True):
    """_tmatrix_disconnected.py:254-289: block-wise over the strongly connected sets."""
    P = np.asarray(P, dtype=np.float64)
    n = P.shape[0]
    dtype = np.float64 if reversible else complex
    R = np.zeros((n, n), dtype=dtype)
    D = np.zeros((n, n), dtype=dtype)
    L = np.zeros((n, n), dtype=dtype)
    for s in connected_sets(P, strong=True):
        I = np.ix_(s, s)
        if len(s) > 1:
            r, d, l = _rdl_block(P[I] / P[I].sum(axis=1)[:, None], reversible)
            R[I], D[I], L[I] = r, np.diag(d), l
        else:
            R[I] = 1
            D[I] = 1
            L[I] = 1
    return R, D, L


def is_reversible(P):
    """_tmatrix_disconnected.py:213-226."""
    P = np.asarray(P, dtype=np.float64)
    for s in connected_sets(P, strong=False):
        Ps = P[s, :][:, s]
        if not is_transition_matrix(Ps):
            return False
        pi = stationary_vector(Ps)
        X = pi[:, None] * Ps
        if not np.allclose(X, X.T):
            return False
    return True


def sample_nonreversible(C, rng=np.random):
    """Non-reversible posterior draw: independent Dirichlet rows,
    P_i ~ Dir(c_i + 1) restricted to positive entries (msmtools sample_tmatrix,
    reversible=False: prior counts -1 ... here the counts already include the prior)."""
    C = np.asarray(C, dtype=np.float64)
    n = C.shape[0]
    P = np.zeros((n, n))
    for i in range(n):
        pos = C[i] > 0
        if np.any(pos):
            P[i, pos] = rng.dirichlet(C[i, pos])
        else:
            P[i, i] = 1.0
    return P


def sample_reversible(C, nsteps=1000, P0=None, rng=np.random):
    """Reversible posterior draw: `nsteps` FULL sweeps (what msmtools' sample_tmatrix(nsteps=...)
    counts) of the element-wise Gibbs sampler of Trendelkamp-Schroer, Wu, Paul, Noe, J. Chem. Phys.
    143, 174101 (2015) on the symmetric flux matrix X (x_ij = pi_i p_ij), prior x_ij^-1, started at
    the reversible MLE.  Diagonal: x_ii / x_i ~ Beta(c_ii, c_i - c_ii) given the rest of the row;
    off-diagonal: independence Metropolis step with a Gamma proposal matched to the conditional
    v^(c0-1) (v + v1)^-c1 (v + v2)^-c2.  This is a plain numpy statement of the same published
    sampler that host_model.cpp's sample_reversible_sweeps implements (which the estimators call);
    it is NOT a line-by-line twin: pairs are visited in lexicographic instead of round-robin order,
    a pair for which no Gamma proposal exists (a == 0 or h >= 0) is skipped where the native code
    takes a log-uniform random-walk step, and there is no accept-at-once branch for v0 == 0 -- same
    stationary distribution, other chain.  `nsteps` full sweeps cost about nsteps * n^2 Python-level
    updates: the native sampler (native_parameters=True, the default) is the one to use beyond a
    few states.  PARITY UNPINNED with respect to msmtools itself (statistical agreement only)."""
    C = np.asarray(C, dtype=np.float64)
    n = C.shape[0]
    if P0 is None:
        P0 = mle_reversible(C + 1e-12)
    pi = stationary_vector(P0)
    X = pi[:, None] * P0
    X = 0.5 * (X + X.T)
    X /= X.sum()
    csum = C.sum(axis=1)
    pos = lambda x: x > 1e-300 and np.isfinite(x)
    for _ in range(int(nsteps)):
        rs = X.sum(axis=1)
        for i in range(n):
            if pos(C[i, i]) and pos(csum[i] - C[i, i]):
                t = rng.beta(C[i, i], csum[i] - C[i, i])
                rest = rs[i] - X[i, i]
                x = t / (1.0 - t) * rest if t < 1.0 else np.inf
                if pos(x):
                    X[i, i] = x
                    rs[i] = rest + x
        for i in range(n):
            for j in range(i):
                c0 = C[i, j] + C[j, i]
                if not c0 > 0:
                    continue
                v0 = X[i, j]
                v1, v2, c1, c2 = rs[i] - v0, rs[j] - v0, csum[i], csum[j]
                a = c1 + c2 - c0
                b = (c1 - c0) * v2 + (c2 - c0) * v1
                c = -c0 * v1 * v2
                vn = v0
                with np.errstate(all='ignore'):
                    vbar = 0.5 * (-b + np.sqrt(b * b - 4.0 * a * c)) / a if a != 0 else np.nan
                    if pos(vbar):
                        h = c1 / (vbar + v1) ** 2 + c2 / (vbar + v2) ** 2 - c0 / vbar ** 2
                        k, itheta = -h * vbar * vbar, -h * vbar
                        if pos(k) and pos(itheta):
                            cand = rng.gamma(k) / itheta
                            if pos(cand):
                                dv = cand - v0
                                dl = ((c0 - k) * np.log1p(dv / v0) - c1 * np.log1p(dv / (v0 + v1))
                                      - c2 * np.log1p(dv / (v0 + v2)) + dv * itheta)
                                if dl >= 0 or rng.random_sample() < np.exp(dl):
                                    vn = cand
                X[i, j] = X[j, i] = vn
                rs[i], rs[j] = v1 + vn, v2 + vn
        X /= X.sum()
    return X / X.sum(axis=1)[:, None]
