"""`pyscf.lib` stand-in: only `param`-style constants some scripts read."""
class param:
    BOHR = 0.52917721092
num_threads = lambda n=None: 1
