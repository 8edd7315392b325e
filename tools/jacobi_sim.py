"""One-sided Jacobi with the XOR tournament order of csrc/bsr_solve.h: solve_cols on (K+1) x K factors of the shape k_solve sees
(siblings' R columns + the candidate's projections): sweeps and rotations from a cold start, from pre-orthogonalised sibling
columns (warm start), and under looser stopping tolerances.  numpy only.  Round 5: warm start 6.2 -> 5.7 sweeps at K = 8, stopping
at 1e-16 instead of 1e-30 6.2 -> 5.8: neither is where k_solve's 19 us at K = 8 are."""
import numpy as np
rs = np.random.RandomState(0)
def jacobi(W, V=None, tol=1e-30, count_rot=False):
    M, K = W.shape
    W = W.copy()
    sweeps = 0; rots = 0
    for sweep in range(40):
        off = 0.0
        for r in range(1, 8):
            done = set()
            for j in range(8):
                p = j ^ r
                if j >= K or p >= K or j in done or p in done or p < j: continue
                done.add(j); done.add(p)
                a, b = j, p
                alpha = W[:, a] @ W[:, a]; beta = W[:, b] @ W[:, b]; gamma = W[:, a] @ W[:, b]
                ab = alpha * beta; g2 = gamma * gamma
                if ab > 0 and g2 > 1e-34 * ab:
                    off = max(off, g2 / ab)
                    zeta = (beta - alpha) / (2 * gamma)
                    t = np.sign(zeta) / (abs(zeta) + np.sqrt(1 + zeta * zeta)) if zeta != 0 else 1.0
                    c = 1 / np.sqrt(1 + t * t); s = c * t
                    wa, wb = W[:, a].copy(), W[:, b].copy()
                    W[:, a] = c * wa - s * wb; W[:, b] = s * wa + c * wb
                    rots += 1
        sweeps += 1
        if off <= tol: break
    return W, sweeps, rots

def trial(K, N=2000, corr=0.5):
    # current columns: correlated features; candidate: another column
    base = rs.standard_normal((N, K + 1))
    mix = np.eye(K + 1) + corr * rs.standard_normal((K + 1, K + 1))
    cols = base @ mix
    X = cols[:, :K]; z = cols[:, K]
    Q, R = np.linalg.qr(X)
    k = rs.randint(K)
    sib = [j for j in range(K) if j != k]
    c = Q.T @ z; w = z - Q @ c; rho = np.linalg.norm(w)
    S = np.zeros((K + 1, K))
    S[:K, :K - 1] = R[:, sib]
    S[:K, K - 1] = c; S[K, K - 1] = rho
    _, cold, rc = jacobi(S)
    # warm: siblings pre-orthogonalised
    W0, _, _ = jacobi(S[:, :K - 1])
    Sw = np.concatenate([W0, S[:, K - 1:]], axis=1)
    _, warm, rw = jacobi(Sw)
    return cold, warm, rc, rw
for K in (3, 5, 8):
    res = np.array([trial(K) for _ in range(300)])
    print("K", K, "cold sweeps mean %.2f  warm sweeps mean %.2f   rotations cold %.1f warm %.1f" % tuple(res.mean(axis=0)))

def trial2(K, tol, N=2000, corr=0.5):
    base = rs.standard_normal((N, K + 1))
    mix = np.eye(K + 1) + corr * rs.standard_normal((K + 1, K + 1))
    cols = base @ mix
    X = cols[:, :K]; z = cols[:, K]
    Q, R = np.linalg.qr(X)
    k = rs.randint(K)
    sib = [j for j in range(K) if j != k]
    c = Q.T @ z; w = z - Q @ c; rho = np.linalg.norm(w)
    S = np.zeros((K + 1, K))
    S[:K, :K - 1] = R[:, sib]
    S[:K, K - 1] = c; S[K, K - 1] = rho
    W, sw, rc = jacobi(S, tol=tol)
    sv = np.sort(np.sqrt((W * W).sum(axis=0)))
    ref = np.sort(np.linalg.svd(S, compute_uv=False))
    return sw, rc, np.max(np.abs(sv - ref) / ref)
for K in (3, 8):
    for tol in (1e-30, 1e-24, 1e-20, 1e-16, 1e-12):
        rs = np.random.RandomState(1)
        res = np.array([trial2(K, tol) for _ in range(200)])
        print("K", K, "tol", tol, "sweeps %.2f rotations %.1f max rel sv err %.2e" % (res[:,0].mean(), res[:,1].mean(), res[:,2].max()))
