"""Minimal `rdkit` STAND-IN (hard import at the top of both in-scope templates,
`templates/calculate_energy.py:13-14`, `templates/optimize_geometry.py:13-14`).  RDKit itself is not
installable here.  Known SMILES map to committed fixture geometries (`mi355scf/smiles_fixtures.py`);
anything else raises with the list of supported strings.  Not a cheminformatics toolkit."""
__version__ = "0.0-mi355x-standin"
from . import Chem  # noqa: F401,E402
