"""`particlesmc params.toml` -- the TOML/CLI surface of src/ParticlesMC.jl:137-299 on the HIP path.

Same keys as the reference; two optional extra keys: `[simulation] mode = "checkerboard" | "sequential"` (default
"checkerboard") selects the device schedule, and `[simulation] rounds = m` (default 1; checkerboard only) lets one grid
shift serve m sweeps (pmc_set_rounds: every cell visit runs m rounds of proposal slots against the neighbourhood it staged).
"""
import os
import sys

import tomli

from .io import EXYZ, LAMMPS, XYZ, load_chains
from .moves import DiscreteSwap, Displacement, DoubleUniform, Move, SimpleGaussian
from .simulation import CALLBACKS, Simulation, build_schedule, run

FORMATS = {"XYZ": XYZ, "EXYZ": EXYZ, "LAMMPS": LAMMPS}


def build_simulation(params, base_dir="."):
    system = params["system"]
    temperature = system["temperature"]
    density = system["density"]
    config = os.path.join(base_dir, system["config"])
    model = system.get("model")
    if model is None:
        model = params["model"]
    if not os.path.isfile(config):
        raise FileNotFoundError(f"Configuration file '{config}' does not exist in the current path.")
    list_type = system.get("list_type", "LinkedList")
    list_parameters = system.get("list_parameters")
    if system.get("bonds") is not None:
        raise NotImplementedError("molecular systems are out of scope of the HIP path")
    sim = params["simulation"]
    steps, burn, seed = sim["steps"], sim.get("burn", 0), sim["seed"]
    output_path = sim.get("output_path", "./")
    mode = sim.get("mode", "checkerboard")
    chains = load_chains(config, args={"temperature": temperature, "density": density, "model": model,
                                       "list_type": list_type, "list_parameters": list_parameters}, seed=seed)
    rounds = int(sim.get("rounds", 1))
    if rounds != 1:
        if mode != "checkerboard":
            raise ValueError("'rounds' applies to mode = \"checkerboard\" only")
        for chain in chains:
            chain.rounds = rounds
    pool = []
    for move in sim["move"]:
        prob, policy, action = move["probability"], move["policy"], move["action"]
        parameters = move.get("parameters", {})
        if action == "Displacement":
            if "sigma" not in parameters:
                raise ValueError(f"Missing parameter 'sigma' for action: {action}")
            if policy != "SimpleGaussian":
                raise ValueError(f"Unsupported policy: {policy} for action: {action}")
            pool.append(Move(Displacement(0, None, 0.0), SimpleGaussian(), {"sigma": parameters["sigma"]}, prob))
        elif action == "DiscreteSwap":
            if "species" not in parameters:
                raise ValueError(f"Missing parameter 'species' for action: {action}")
            sp = parameters["species"]
            if len(sp) != 2 or not all(isinstance(s, int) for s in sp):
                raise ValueError(f"'species' for action: {action} must be an array of two ints")
            if policy != "DoubleUniform":
                raise ValueError(f"Unsupported policy: {policy} for action: {action}")
            pool.append(Move(DiscreteSwap(sp, chains[0]), DoubleUniform(), [], prob))
        else:
            raise ValueError(f"Unsupported action: {action}")
    algorithm_list = [dict(algorithm="Metropolis", pool=pool, seed=seed, parallel=sim["parallel"],
                           sweepstep=len(chains[0]), mode=mode)]
    for output in sim.get("output", []):
        alg = output["algorithm"]
        sp = output["scheduler_params"]
        interval = sp["linear_interval"]
        if "log_base" in sp:
            block = sorted(set([0] + [int(2.0 ** k) for k in range(0, 64) if 2.0 ** k <= interval]))
            sched = build_schedule(steps, burn, block)
        else:
            sched = build_schedule(steps, burn, interval)
        if alg == "StoreCallbacks":
            algorithm_list.append(dict(algorithm=alg, callbacks=[CALLBACKS[c] for c in output.get("callbacks", [])],
                                       scheduler=sched))
        elif alg == "StoreAcceptance":
            algorithm_list.append(dict(algorithm=alg, dependencies=output.get("dependencies"), scheduler=sched))
        elif alg in ("StoreTrajectories", "StoreLastFrames"):
            algorithm_list.append(dict(algorithm=alg, scheduler=sched, fmt=FORMATS[output.get("fmt", "XYZ")]()))
        elif alg == "PrintTimeSteps":
            algorithm_list.append(dict(algorithm=alg, scheduler=sched))
        else:
            raise ValueError(f"Unsupported output algorithm: {alg}")
    return Simulation(chains, algorithm_list, steps, path=os.path.join(base_dir, output_path), verbose=True)


def particlesmc(params_path):
    if not os.path.isfile(params_path):
        raise FileNotFoundError(f"Parameter file '{params_path}' does not exist in the current path.")
    with open(params_path, "rb") as f:
        params = tomli.load(f)
    sim = build_simulation(params, base_dir=os.path.dirname(os.path.abspath(params_path)))
    return run(sim)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        print("usage: particlesmc <params.toml>", file=sys.stderr)
        return 2
    particlesmc(argv[0])
    return 0


if __name__ == "__main__":
    sys.exit(main())
