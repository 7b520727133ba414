"""particlesmc_amd -- MI355X-native (gfx950) implementation of the ParticlesMC
pair-energy + Metropolis hot path, behind the reference's move/policy API.

All compute runs in csrc/libpmc_hip.so through the C ABI of include/pmc.h; this
package is the host-side mirror of the reference interface (src/ParticlesMC.jl:115-127).
"""
from ._lib import PMCError, device_count, lib  # noqa: F401
from .models import (BHHP, JBB, KobAndersen, LennardJones, Model, SmoothLennardJones,  # noqa: F401
                     SoftSpheres, cutoff, cutoff2, potential)
from .neighbours import CellList, EmptyList, LinkedList, NeighbourList, VerletList  # noqa: F401
from .system import (Atoms, System, compute_energy_particle, compute_energy_particles,
                     compute_energy_particles_exact, fold_back,  # noqa: F401
                     get_box, get_model, get_neighbour_list, get_position, get_species)
from .moves import (DiscreteSwap, Displacement, DoubleUniform, EnergyBias, Move, SequentialStream, SimpleGaussian,  # noqa: F401
                    delta_log_target_density, invert_action, log_proposal_density, mc_step,
                    perform_action, revert_action, reward, sample_action, unnormalised_log_target_density)
from .simulation import Metropolis, Simulation, build_schedule, energy, mc_sweep, mc_sweep_many, run  # noqa: F401
from .io import EXYZ, LAMMPS, XYZ, load_chains, load_configuration  # noqa: F401
