#!/usr/bin/env python3
"""Writes mi355scf/data/lebedev.npz: Lebedev-Laikov sphere rules (unit vectors x_<n>, weights w_<n>
summing to 1) for the point counts PySCF-style grids use.  Source: scipy.integrate.lebedev_rule
(scipy 1.15.3).  The product reads the committed file and never imports scipy at run time."""
import os
import numpy as np
from scipy.integrate import lebedev_rule

DEGREE = {6: 3, 14: 5, 26: 7, 38: 9, 50: 11, 74: 13, 86: 15, 110: 17, 146: 19, 170: 21, 194: 23, 230: 25,
          266: 27, 302: 29, 350: 31, 434: 35, 590: 41, 770: 47, 974: 53}
out = {}
for n, deg in DEGREE.items():
    x, w = lebedev_rule(deg)
    assert x.shape == (3, n)
    out[f"x_{n}"] = np.ascontiguousarray(x.T)
    out[f"w_{n}"] = w / (4 * np.pi)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "computational-chemistry-ai_amd", "python", "mi355scf", "data", "lebedev.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes")
