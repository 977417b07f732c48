from mi355scf.infrared import Infrared  # noqa: F401
Infrared = Infrared
