"""Redundant primitive internal coordinates for the geometry optimiser (SURVEY.md section 8f rank 1:
"internal-coordinate BFGS"; the reference reaches this through geomeTRIC, `templates/optimize_geometry.py:99`).

Host-side NumPy only (3N <= a few hundred): bond graph from covalent radii, stretches / bends / torsions /
out-of-plane torsions, analytic Wilson B rows, generalised inverse of G = B B^T, and the iterative
back-transformation of an internal-coordinate step to Cartesians [Peng, Ayala, Schlegel, Frisch, JCC 17, 49
(1996); Bakken, Helgaker, JCP 117, 9160 (2002)].
"""
import itertools
import os

import numpy as np

BOHR = 0.52917721092
_COV = {1: 0.31, 2: 0.28, 3: 1.28, 4: 0.96, 5: 0.84, 6: 0.76, 7: 0.71, 8: 0.66, 9: 0.57, 10: 0.58,
        11: 1.66, 12: 1.41, 13: 1.21, 14: 1.11, 15: 1.07, 16: 1.05, 17: 1.02, 18: 1.06}
LINEAR = np.deg2rad(170.0)


def bond_graph(z, x, scale=1.3):
    """Bonds (i > j) closer than scale * (r_cov_i + r_cov_j); fragments are joined by their closest contacts."""
    n = len(x)
    rc = np.array([_COV.get(int(q), 1.2) for q in z]) / BOHR
    d = np.linalg.norm(x[:, None, :] - x[None, :, :], axis=2)
    bonds = [(i, j) for i in range(n) for j in range(i) if d[i, j] < scale * (rc[i] + rc[j])]
    # union-find over fragments; connect the closest pair of atoms of different fragments until one fragment is left
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for i, j in bonds:
        parent[find(i)] = find(j)
    while len({find(a) for a in range(n)}) > 1:
        best = None
        for i in range(n):
            for j in range(i):
                if find(i) != find(j) and (best is None or d[i, j] < best[0]):
                    best = (d[i, j], i, j)
        _, i, j = best
        bonds.append((i, j))
        parent[find(i)] = find(j)
    return bonds


def _angle_value(x, i, j, k):
    a, b = x[i] - x[j], x[k] - x[j]
    return np.arccos(np.clip(a @ b / np.linalg.norm(a) / np.linalg.norm(b), -1.0, 1.0))


def _dihedral_value(x, i, j, k, l):
    b0, b1, b2 = x[i] - x[j], x[k] - x[j], x[l] - x[k]
    b1n = b1 / np.linalg.norm(b1)
    v, w = b0 - (b0 @ b1n) * b1n, b2 - (b2 @ b1n) * b1n
    return np.arctan2(np.cross(b1n, v) @ w, v @ w)


class Internals:
    """Primitive set for one molecule.  `kinds[q]` in {"bond","angle","dihedral"}, `atoms[q]` the index tuple."""

    def __init__(self, z, x):
        n = len(x)
        self.natm = n
        self.z = np.asarray(z, dtype=int)
        bonds = bond_graph(z, x)
        nb = [[] for _ in range(n)]
        for i, j in bonds:
            nb[i].append(j); nb[j].append(i)
        self.kinds, self.atoms = [], []
        for b in bonds:
            self.kinds.append("bond"); self.atoms.append(b)
        self.has_linear = False
        self.oop = set()            # indices of the out-of-plane torsions (their middle pair is not bonded)
        for j in range(n):
            for i, k in itertools.combinations(sorted(nb[j]), 2):
                if _angle_value(x, i, j, k) > LINEAR:
                    self.has_linear = True
                    continue
                self.kinds.append("angle"); self.atoms.append((i, j, k))
        seen = set()
        for (j, k) in bonds:
            for i in nb[j]:
                for l in nb[k]:
                    if len({i, j, k, l}) != 4 or (l, k, j, i) in seen:
                        continue
                    if _angle_value(x, i, j, k) > LINEAR or _angle_value(x, j, k, l) > LINEAR:
                        continue
                    seen.add((i, j, k, l))
                    self.kinds.append("dihedral"); self.atoms.append((i, j, k, l))
        # out-of-plane motion of three-coordinate centres (e.g. the carbonyl carbon of H2CO): torsion c-a-b-d
        for c in range(n):
            if len(nb[c]) == 3:
                a, b, d_ = sorted(nb[c])
                if min(_angle_value(x, c, a, b), _angle_value(x, a, b, d_)) > np.deg2rad(5.0) and \
                        max(_angle_value(x, c, a, b), _angle_value(x, a, b, d_)) < LINEAR:
                    self.kinds.append("dihedral"); self.atoms.append((c, a, b, d_))
                    self.oop.add(len(self.kinds) - 1)
        self.nq = len(self.kinds)

    # ---------------------------------------------------------------------------------------------
    def values(self, x):
        q = np.empty(self.nq)
        for n_, (kind, idx) in enumerate(zip(self.kinds, self.atoms)):
            if kind == "bond":
                q[n_] = np.linalg.norm(x[idx[0]] - x[idx[1]])
            elif kind == "angle":
                q[n_] = _angle_value(x, *idx)
            else:
                q[n_] = _dihedral_value(x, *idx)
        return q

    def diff(self, q1, q0):
        """q1 - q0 with torsions wrapped into (-pi, pi]."""
        d = q1 - q0
        for n_, kind in enumerate(self.kinds):
            if kind == "dihedral":
                d[n_] = (d[n_] + np.pi) % (2.0 * np.pi) - np.pi
        return d

    def bmatrix(self, x):
        """Wilson B matrix dq/dx [nq, 3N] (analytic rows)."""
        B = np.zeros((self.nq, 3 * self.natm))
        for n_, (kind, idx) in enumerate(zip(self.kinds, self.atoms)):
            row = B[n_].reshape(-1, 3)
            if kind == "bond":
                i, j = idx
                u = x[i] - x[j]
                u /= np.linalg.norm(u)
                row[i] += u; row[j] -= u
            elif kind == "angle":
                i, j, k = idx
                u, v = x[i] - x[j], x[k] - x[j]
                lu, lv = np.linalg.norm(u), np.linalg.norm(v)
                u, v = u / lu, v / lv
                c = np.clip(u @ v, -1.0, 1.0)
                s = np.sqrt(max(1.0 - c * c, 1e-12))
                di = (c * u - v) / (lu * s)
                dk = (c * v - u) / (lv * s)
                row[i] += di; row[k] += dk; row[j] -= di + dk
            else:
                i, j, k, l = idx
                # torsion i-j-k-l (Blondel & Karplus form)
                F, G, H = x[i] - x[j], x[j] - x[k], x[l] - x[k]
                A, Bv = np.cross(F, G), np.cross(H, G)
                lG = np.linalg.norm(G)
                A2, B2 = A @ A, Bv @ Bv
                di = -lG / A2 * A
                dl = lG / B2 * Bv
                fg, hg = F @ G, H @ G
                dj = lG / A2 * A + fg / (A2 * lG) * A - hg / (B2 * lG) * Bv
                dk = -lG / B2 * Bv - fg / (A2 * lG) * A + hg / (B2 * lG) * Bv
                row[i] += di; row[j] += dj; row[k] += dk; row[l] += dl
        return B

    HESS_MODEL = "lindh"

    def guess_hessian_diag(self, x=None, model=None):
        """Diagonal guess (a.u.) for the primitives.
        "lindh" (default): Lindh, Bernhardsson, Karlstrom, Malmqvist, CPL 241, 423 (1995) -- k = 0.45 rho_ij (stretch),
            0.15 rho_ij rho_jk (bend), 0.005 rho_ij rho_jk rho_kl (torsion), rho_ij = exp(alpha_ij (r_ref,ij^2 - r_ij^2)) with
            alpha / r_ref tabulated per pair of periods: soft torsions, so that the many redundant torsions about one rotatable
            bond do not add up to a rotor forty times stiffer than it is (the round-1 constants did, and the optimisation of
            ibuprofen crawled along its methyl / isobutyl rotors for ten steps);
        "geometric": the constants geomeTRIC [MEM] starts from (0.35 / 0.16 / 0.023);
        "simple": round 1 (0.5 / 0.2 / 0.1)."""
        model = (model or os.environ.get("MI355_OPT_HESS") or self.HESS_MODEL).lower()
        if model == "simple":
            return np.array([{"bond": 0.5, "angle": 0.2, "dihedral": 0.1}[k] for k in self.kinds])
        if model == "geometric" or x is None:
            return np.array([{"bond": 0.35, "angle": 0.16, "dihedral": 0.023}[k] for k in self.kinds])
        per = np.where(self.z <= 2, 0, np.where(self.z <= 10, 1, 2))
        alpha = np.array([[1.0000, 0.3949, 0.3949], [0.3949, 0.2800, 0.2800], [0.3949, 0.2800, 0.2800]])
        rref = np.array([[1.35, 2.10, 2.53], [2.10, 2.87, 3.40], [2.53, 3.40, 3.40]])

        def rho(i, j):
            r2 = float(((x[i] - x[j]) ** 2).sum())
            return np.exp(alpha[per[i], per[j]] * (rref[per[i], per[j]] ** 2 - r2))
        out = []
        for q, (k, a) in enumerate(zip(self.kinds, self.atoms)):
            if q in self.oop:     # centre c = a[0] with neighbours a[1:]: geomeTRIC's out-of-plane constant, Lindh-damped bonds
                out.append(0.045 * rho(a[0], a[1]) * rho(a[0], a[2]) * rho(a[0], a[3]))
            elif k == "bond":
                out.append(0.45 * rho(a[0], a[1]))
            elif k == "angle":
                out.append(0.15 * rho(a[0], a[1]) * rho(a[1], a[2]))
            else:
                out.append(0.005 * rho(a[0], a[1]) * rho(a[1], a[2]) * rho(a[2], a[3]))
        return np.maximum(np.array(out), 1e-3)

    @staticmethod
    def ginv(B):
        """Generalised inverse of G = B B^T and the projector P = G G^- onto the non-redundant space."""
        G = B @ B.T
        w, v = np.linalg.eigh(G)
        keep = w > 1e-8 * max(w.max(), 1e-30)
        Ginv = (v[:, keep] / w[keep]) @ v[:, keep].T
        return Ginv, G @ Ginv, int(keep.sum())

    def to_cartesian(self, x0, dq, maxit=40):
        """Cartesian geometry whose internal coordinates are q(x0) + dq (iterative back-transformation).
        Returns (x, achieved dq)."""
        x = x0.copy()
        q0 = self.values(x0)
        target = dq.copy()
        first = None
        best = (np.inf, None)
        for it in range(maxit):
            B = self.bmatrix(x)
            Ginv, _, _ = self.ginv(B)
            resid = target - self.diff(self.values(x), q0)
            dx = (B.T @ (Ginv @ resid)).reshape(-1, 3)
            x = x + dx
            if first is None:
                first = x.copy()
            err = np.sqrt((dx ** 2).mean())
            if err < best[0]:
                best = (err, x.copy())
            if err < 1e-10:
                break
            if it > 5 and err > 10.0 * best[0]:
                break  # diverging: take the first-order step
        else:
            it = maxit
        x = best[1] if best[0] < 1e-6 else first
        return x, self.diff(self.values(x), q0)
