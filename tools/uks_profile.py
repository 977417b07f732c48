#!/usr/bin/env python3
"""cProfile of the second kernel() of a UKS object (benzene cation B3LYP/cc-pVTZ): the plain UHF/UKS loop still syncs ~9 times per
cycle (`.cpu()` reads), unlike the single-readback RHF/RKS step.   python tools/uks_profile.py"""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "computational-chemistry-ai_amd", "python"))
import torch
from mi355scf.mole import Mole
from mi355scf.uks import UKS
from mi355scf import fixtures
mol = Mole(atom=fixtures.BENZENE, basis="cc-pVTZ", verbose=0, charge=1, spin=1).build()
mf = UKS(mol); mf.xc = "B3LYP"; mf.conv_tol = 1e-9
mf.kernel()
pr = cProfile.Profile(); pr.enable(); mf.kernel(); torch.cuda.synchronize(); pr.disable()
print("cycles", mf.cycles, mf.timing)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
