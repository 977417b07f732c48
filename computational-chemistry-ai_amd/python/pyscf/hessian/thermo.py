from mi355scf.thermo import harmonic_analysis, thermo, rotational_symmetry_number  # noqa: F401
