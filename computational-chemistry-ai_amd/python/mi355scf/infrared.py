"""`pyscf.prop.infrared.{RHF,RKS}` (call sites `templates/calculate_ir_spectrum.py:17,90-105`, `templates/opt-freq.py:17,
425-447`): harmonic frequencies and IR intensities.  `kernel()` runs the semi-numerical Hessian (`hessian.py`), whose
displaced SCF solutions also give the dipole derivatives d mu / d R; intensities are
I_k = (N_A pi / 3 c^2) |d mu / d Q_k|^2 with Q the mass-weighted normal coordinates: 974.8801 km/mol per (e^2 / amu).
"""
import numpy as np

from . import thermo
from .hessian import Hessian

KM_PER_MOL = 974.8801   # (e^2/amu) -> km/mol  [= 42.2561 km/mol per (D/A)^2/amu * (4.80320 D/A per e)^2]


class Infrared:
    def __init__(self, mf):
        self.base = mf
        self.mol = mf.mol
        self.verbose = mf.verbose
        self.hessian = None          # may be preset (opt-freq.py:438); the displaced SCFs still run for d mu / d R
        self.freq_info = None
        self.ir_intensity = None
        self.dipole_deriv = None

    def kernel(self):
        h = Hessian(self.base)
        hess = h.kernel()
        if self.hessian is None:
            self.hessian = hess
        self.dipole_deriv = h.dipole_deriv
        mol = self.base.mol
        info = thermo.harmonic_analysis(mol, self.hessian)
        mass = mol.atom_mass_list(isotope_avg=True)
        n = mol.natm
        # mass-weighted orthonormal modes from the normalised Cartesian ones: L_ik = x_ik sqrt(m_i) / |...|
        modes = info["norm_mode"] * np.sqrt(mass)[None, :, None]
        modes = modes / np.sqrt(np.einsum("kix,kix->k", modes, modes))[:, None, None]
        dmu_dq = np.einsum("ixc,kix->kc", self.dipole_deriv / np.sqrt(np.where(mass > 0, mass, 1.0))[:, None, None], modes)
        self.ir_intensity = KM_PER_MOL * np.einsum("kc,kc->k", dmu_dq, dmu_dq)
        self.freq_info = info
        self.vib_dict = info
        return info

    def summary(self):
        if self.freq_info is None:
            self.kernel()
        log = lambda m: self.base._log(3, m)
        log("  mode   frequency (cm^-1)   IR intensity (km/mol)")
        for k, (f, i) in enumerate(zip(self.freq_info["freq_wavenumber"], self.ir_intensity)):
            log(f"  {k + 1:4d}   {f:14.2f}   {i:14.3f}")
        return self


RHF = RKS = UHF = UKS = Infrared
