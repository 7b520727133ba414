"""Moves and policies: host mirror of src/moves.jl and src/utils.jl:4-10 (Arianna verbs).

The per-move verbs are the batch-of-one form of the device path; throughput
comes from `mc_sweep` (simulation.py), the analogue of Arianna's mc_sweep!.
"""
import ctypes as C
import math

import numpy as np

from . import _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Action:
    pass


class Policy:
    pass


class Displacement(Action):
    """mutable struct Displacement(i, delta, de) -- src/moves.jl:34-38."""

    def __init__(self, i=0, delta=None, de=0.0):
        self.i = int(i)
        self.delta = np.zeros(3) if delta is None else np.array(delta, dtype=np.float64)
        self.de = float(de)


class DiscreteSwap(Action):
    """mutable struct DiscreteSwap(i, j, species, particles_per_species, de) -- src/moves.jl:137-149."""

    def __init__(self, *args):
        if len(args) == 2:                      # DiscreteSwap(species::Vector{Int}, system)
            species, system = args
            sp = system.species
            self.i, self.j = 1, 1
            self.species = (int(species[0]), int(species[1]))
            self.particles_per_species = (int(np.count_nonzero(sp == species[0])),
                                          int(np.count_nonzero(sp == species[1])))
            self.de = 0.0
        else:
            i, j, species, pps, de = args
            self.i, self.j = int(i), int(j)
            self.species = (int(species[0]), int(species[1]))
            self.particles_per_species = (int(pps[0]), int(pps[1]))
            self.de = float(de)


class SimpleGaussian(Policy):
    """src/moves.jl:105."""


class DoubleUniform(Policy):
    """src/moves.jl:226."""


class EnergyBias(Policy):
    """src/moves.jl:246 -- pick the swap pair with probabilities ~ exp(theta_k * e_particle); needs the
    per-particle energies of both species for every proposal (sequential chain only)."""


class Move:
    """Arianna Move(action, policy, parameters, probability)."""

    def __init__(self, action, policy, parameters, probability):
        self.action, self.policy, self.parameters, self.probability = action, policy, parameters, float(probability)

    @property
    def sigma(self):
        p = self.parameters
        if isinstance(p, dict):
            return float(p.get("sigma", p.get("σ", 0.0)))
        return float(getattr(p, "sigma", 0.0)) if p is not None and not isinstance(p, (list, tuple)) else 0.0


# ---- verbs -------------------------------------------------------------------

def perform_action(system, action):
    """Arianna.perform_action!(system, action) -- src/moves.jl:11-20 over 57-67 / 187-194.

    Returns (e1, e2); sets action.de and updates system.energy[1] on the device."""
    e1, e2, de = C.c_double(), C.c_double(), C.c_double()
    L = _lib.lib()
    if isinstance(action, Displacement):
        d = np.ascontiguousarray(action.delta[: system.d], dtype=np.float64)
        _lib.check(L.pmc_displacement_perform(system._h, action.i, _dp(d), C.byref(e1), C.byref(e2), C.byref(de)),
                   system._h)
    elif isinstance(action, DiscreteSwap):
        _lib.check(L.pmc_swap_perform(system._h, action.i, action.j, C.byref(e1), C.byref(e2), C.byref(de)),
                   system._h)
    else:
        raise TypeError(f"unsupported action {type(action).__name__}")
    action.de = de.value
    return e1.value, e2.value


def invert_action(action, system=None):
    """src/moves.jl:88-90, 212-214."""
    if isinstance(action, Displacement):
        action.delta = -action.delta
    else:
        action.i, action.j = action.j, action.i


def revert_action(system, action):
    """Arianna.revert_action! -- src/moves.jl:76-81, 201-207 (call after invert_action)."""
    L = _lib.lib()
    if isinstance(action, Displacement):
        d = np.ascontiguousarray(action.delta[: system.d], dtype=np.float64)
        _lib.check(L.pmc_displacement_revert(system._h, action.i, _dp(d), action.de), system._h)
    else:
        _lib.check(L.pmc_swap_revert(system._h, action.i, action.j, action.de), system._h)


def _theta(parameters):
    if isinstance(parameters, dict):
        return float(parameters.get("theta1", parameters.get("θ₁"))), float(parameters.get("theta2", parameters.get("θ₂")))
    return float(parameters[0]), float(parameters[1])


def energy_bias_sample(system, species, parameters, u1, u2):
    """pmc_energy_bias_sample: the two energy-weighted picks and log q(i, j) on the device (src/moves.jl:254-279);
    only the pair and one double come back."""
    th1, th2 = _theta(parameters)
    i, j, lq = C.c_int64(), C.c_int64(), C.c_double()
    _lib.check(_lib.lib().pmc_energy_bias_sample(system._h, int(species[0]), int(species[1]), th1, th2, float(u1), float(u2),
                                                 C.byref(i), C.byref(j), C.byref(lq)), system._h)
    return i.value, j.value, lq.value


def energy_bias_logq(system, species, parameters, i, j):
    """pmc_energy_bias_logq: log_proposal_density(::DiscreteSwap, ::EnergyBias, ...) of the pair (i, j) in the current state."""
    th1, th2 = _theta(parameters)
    lq = C.c_double()
    _lib.check(_lib.lib().pmc_energy_bias_logq(system._h, int(species[0]), int(species[1]), th1, th2, int(i), int(j),
                                               C.byref(lq)), system._h)
    return lq.value


def log_proposal_density(action, policy, parameters, system):
    """src/moves.jl:110-112, 231-233, 254-263."""
    if isinstance(policy, EnergyBias):
        return energy_bias_logq(system, action.species, parameters, action.i, action.j)
    if isinstance(action, Displacement):
        sigma = parameters["sigma"] if isinstance(parameters, dict) else float(parameters)
        n2 = float(np.dot(action.delta[: system.d], action.delta[: system.d]))
        return -n2 / (2 * sigma ** 2) - system.d * math.log(2 * math.pi * sigma ** 2) / 2
    return -math.log(action.particles_per_species[0] * action.particles_per_species[1])


def delta_log_target_density(e1, e2, system):
    """src/utils.jl:8-10."""
    return -(e2 - e1) / system.temperature


def unnormalised_log_target_density(e, system):
    """src/utils.jl:4-6."""
    return -e / system.temperature


def reward(action, system):
    """Arianna.PolicyGuided.reward -- src/moves.jl:98-100, 219-221."""
    if isinstance(action, Displacement):
        return float(np.dot(action.delta[: system.d], action.delta[: system.d]))
    return 1.0


class SequentialStream:
    """The Philox stream the device's sequential chain consumes (DESIGN.md "random
    streams"): one (R, Q) word pair per step.  Driving sample_action / mc_step from
    it reproduces pmc_sweep(PMC_SEQUENTIAL) move for move."""

    def __init__(self, seed, chain=0, step=0):
        self.seed, self.chain, self.step = int(seed), int(chain), int(step)
        self._load()

    def _load(self):
        R = (C.c_uint32 * 4)()
        Q = (C.c_uint32 * 4)()
        _lib.check(_lib.lib().pmc_sequential_words(self.seed, self.chain, self.step, R, Q))
        self.R, self.Q = list(R), list(Q)

    def advance(self):
        self.step += 1
        self._load()

    @staticmethod
    def mulhi(w, n):
        return (w * n) >> 32

    def normals(self):
        z = [C.c_double() for _ in range(4)]
        _lib.lib().pmc_box_muller(self.Q[0], self.Q[1], C.byref(z[0]), C.byref(z[1]))
        _lib.lib().pmc_box_muller(self.Q[2], self.Q[3], C.byref(z[2]), C.byref(z[3]))
        return [v.value for v in z]

    def log_uniform(self):
        v = C.c_double()
        _lib.lib().pmc_log_uniform(self.R[2], self.R[3], C.byref(v))
        return v.value

    def select_move(self, pool):
        if len(pool) == 1:
            return 0
        tot = 0.0
        for mv in pool:
            tot += mv.probability
        u = self.R[0] * (1.0 / 4294967296.0) * tot
        acc = 0.0
        for m, mv in enumerate(pool):
            acc += mv.probability
            if u < acc:
                return m
        return len(pool) - 1


def sample_action(action, policy, parameters, system, rng):
    """Arianna.sample_action! -- src/moves.jl:120-123, 238-241; rng is a SequentialStream."""
    if isinstance(policy, EnergyBias):
        # rand(rng, Categorical(w / sum(w))) for both species (moves.jl:274-279): inverse-CDF picks on the device from
        # the step's two 32-bit words
        u1, u2 = rng.R[1] * (1.0 / 4294967296.0), rng.Q[0] * (1.0 / 4294967296.0)
        action.i, action.j, _ = energy_bias_sample(system, action.species, parameters, u1, u2)
        return
    if isinstance(action, Displacement):
        sigma = parameters["sigma"] if isinstance(parameters, dict) else float(parameters)
        action.i = 1 + rng.mulhi(rng.R[1], system.N)
        z = rng.normals()
        action.delta = np.array([sigma * z[a] for a in range(system.d)])
    else:
        head_ids = system._species_ids_device(action.species)
        ids_a, ids_b = head_ids
        action.i = int(ids_a[rng.mulhi(rng.R[1], len(ids_a))])
        action.j = int(ids_b[rng.mulhi(rng.Q[0], len(ids_b))])


def mc_step(system, action, policy, parameters, rng):
    """Arianna mc_step! call contract on the per-move verbs.  Returns accepted?"""
    sample_action(action, policy, parameters, system, rng)
    logq_f = log_proposal_density(action, policy, parameters, system)
    e1, e2 = perform_action(system, action)
    dlp = delta_log_target_density(e1, e2, system)
    invert_action(action, system)
    logq_b = log_proposal_density(action, policy, parameters, system)
    ok = rng.log_uniform() < dlp + (logq_b - logq_f)
    if not ok:
        revert_action(system, action)
    return ok
