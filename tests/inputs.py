"""Seeded synthetic inputs of the parity tests: the generators live in the package
(particlesmc_amd/workloads.py) so that bench.py and smoke() do not import the test tree."""
from particlesmc_amd.workloads import binary_lattice, lj_lattice, random_gas  # noqa: F401
