"""`pyscf.scf.rohf`: the class exists for `isinstance(mf, (scf.rhf.RHF, scf.rohf.ROHF))` (`templates/calculate_reaction_energy.py:167`);
restricted open-shell SCF itself is not implemented on the MI355X engine."""


class ROHF:
    def __init__(self, *a, **k):
        raise NotImplementedError("ROHF is not implemented in the MI355X engine (use scf.UHF / dft.UKS for open shells)")
