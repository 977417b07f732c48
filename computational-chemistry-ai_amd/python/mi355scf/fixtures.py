"""Committed geometry fixtures for the BASELINE configs, Angstrom.

The reference obtains geometries from RDKit ETKDG+MMFF (`templates/calculate_energy.py:62-81`), which is
absent here; these fixed geometries define the synthetic benchmark inputs instead."""
import math


def _benzene(rcc=1.3915, rch=1.0800):
    lines = []
    for k in range(6):
        a = math.pi / 3 * k
        lines.append(f"C {rcc * math.cos(a):.8f} {rcc * math.sin(a):.8f} 0.0")
    for k in range(6):
        a = math.pi / 3 * k
        r = rcc + rch
        lines.append(f"H {r * math.cos(a):.8f} {r * math.sin(a):.8f} 0.0")
    return "; ".join(lines)


BENZENE = _benzene()
H2CO = "C 0.0 0.0 0.0; O 1.2 0.0 0.0; H -0.5 0.9 0.0; H -0.5 -0.9 0.0"  # reference README.md:187-192
H2O = "O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587"
