"""Configuration readers / writers either side of the hot path: host mirror of src/IO/.

XYZ (src/IO/xyz.jl), EXYZ (src/IO/exyz.jl), LAMMPS (src/IO/lammps.jl) for atomic
systems, and load_chains (src/IO/IO.jl:210-331).  Molecules/bonds are out of scope.
"""
import os
import re

import numpy as np

from . import models as M
from .neighbours import LIST_TYPES, LinkedList
from .system import System, fold_back


class Format:
    extension = ""


class XYZ(Format):
    extension = ".xyz"


class EXYZ(Format):
    extension = ".exyz"


class LAMMPS(Format):
    extension = ".lammpstrj"


def _format_of(filename):
    if filename.endswith(".xyz"):
        return XYZ()
    if filename.endswith(".exyz"):
        return EXYZ()
    if filename.endswith((".lmp", ".lammpstrj", ".lammps")):
        return LAMMPS()
    raise ValueError(f"Unsupported file format: {filename}")        # IO.jl:37


def _read_header(data, fmt):
    if isinstance(fmt, XYZ):                                          # xyz.jl:39-51
        N = int(data[0])
        metadata = data[1].split(" ")
        cell = next(m for m in metadata if m.startswith("cell:"))[5:]
        box = np.array([float(v) for v in cell.split(",")])
        cols = next(m for m in metadata if m.startswith("columns:"))[8:].split(",")
        info, idx = {}, 0
        for name in cols:
            if name == "species":
                info["species"] = (1, idx)
            elif name == "position":
                info["pos"] = (len(box), idx)
            elif name == "molecule":
                raise NotImplementedError("molecular configurations are out of scope")
            else:
                raise ValueError(f"{name} is not supported")
            idx += 1
        return N, box, info, metadata, 2
    if isinstance(fmt, EXYZ):                                         # exyz.jl:28-48
        N = int(data[0])
        line = data[1]
        mat = re.search(r'Lattice="(.*?)"', line)
        if mat is None:
            raise ValueError("Invalid Lattice line format")
        vals = [float(v) for v in mat.group(1).split()]
        if len(vals) != 9:
            raise ValueError("Lattice matrix must have 9 elements")
        box = np.array([vals[0], vals[4], vals[8]])
        cols = re.search(r"Properties=(.*)", line).group(1).split(":")
        info, i, idx = {}, 0, 0
        while i < len(cols):
            if i + 2 < len(cols) + 0 and cols[i + 1] in ("S", "I", "R"):
                dim = int(cols[i + 2].split()[0])
                info[cols[i]] = (dim, idx)
                idx += dim
                i += 3
            else:
                i += 1
        return N, box, info, line.split(" "), 2
    # LAMMPS, lammps.jl:75-86
    ia = next(k for k, l in enumerate(data) if "ITEM: NUMBER OF ATOMS" in l)
    ib = next(k for k, l in enumerate(data) if "ITEM: BOX BOUNDS" in l)
    ic = next(k for k, l in enumerate(data) if "ITEM: ATOMS" in l)
    N = int(data[ia + 1])
    bounds = [[float(v) for v in data[ib + 1 + k].split()] for k in range(3)]
    box = np.array([b[1] - b[0] for b in bounds])
    cols = data[ic].split(" ")
    info, idx = {}, 0
    for name in cols:
        if name in ("ITEM:", "ATOMS", "y", "z"):
            if name == "z":
                pass
            continue
        if name == "type":
            info["species"] = (1, idx)
        elif name == "x":
            info["pos"] = (3 if "z" in cols else 2, idx)
        elif name == "":
            continue
        else:
            raise ValueError(f"{name} is not supported")
        idx += 1
    return N, box, info, [], ic + 1


def load_configuration(filename, m=1):
    """load_configuration(filename) -- src/IO/IO.jl:27-100 (frame m = 1)."""
    fmt = _format_of(filename)
    with open(filename) as f:
        data = f.read().split("\n")
    N, box, info, metadata, row0 = _read_header(data, fmt)
    if "pos" not in info:
        raise KeyError("pos array has not been found in metadata")
    pos_d, pos_i = info["pos"]
    if pos_d < len(box):
        box = box[:pos_d]                                             # IO.jl:75-77
    species = np.ones(N, dtype=np.int32)
    position = np.zeros((N, pos_d))
    for k in range(N):
        parts = data[row0 + k].split(" ")
        if "species" in info:
            species[k] = int(parts[info["species"][1]])
        position[k] = [float(v) for v in parts[pos_i: pos_i + pos_d]]
    return {"N": N, "d": pos_d, "box": box, "species": species, "position": position, "metadata": metadata}


def _volume_sphere(r, d):                                             # utils.jl:55-59
    if d == 0:
        return 1.0
    if d == 1:
        return 2 * r
    return 2 * np.pi * r ** 2 * _volume_sphere(r, d - 2) / d


def parse_model(spec, n_species):
    """Model selection of load_chains -- src/IO/IO.jl:296-306."""
    if isinstance(spec, dict):
        def key(i, j):
            return f"{i}-{j}" if i <= j else f"{j}-{i}"
        return [[M.model_from_dict(spec[key(i, j)]) for j in range(1, n_species + 1)] for i in range(1, n_species + 1)]
    name = spec.strip()
    if name.endswith("()"):
        name = name[:-2]
    if name in M.PRESETS:
        return M.PRESETS[name]()
    raise ValueError(f"unknown model preset {spec!r} (the reference evals arbitrary Julia here; presets only)")


def load_chains(init_path, args=None, filename="", verbose=False, seed=0, device=0):
    """load_chains(init_path; args) -- src/IO/IO.jl:210-331 for atomic systems."""
    args = dict(args or {})
    files = []
    if os.path.isfile(init_path):
        files.append(init_path)
    elif os.path.isdir(init_path):
        for root, _, fs in os.walk(init_path):
            files += [os.path.join(root, f) for f in fs if filename in f]
    cfgs = [load_configuration(f) for f in files]
    if not cfgs:
        raise FileNotFoundError(init_path)
    N, d = cfgs[0]["N"], cfgs[0]["d"]
    positions = [c["position"] for c in cfgs]
    boxes = [np.array(c["box"], dtype=np.float64) for c in cfgs]
    species = [c["species"] for c in cfgs]
    densities = [len(p) / np.prod(b) for p, b in zip(positions, boxes)]            # IO.jl:234
    temps = None
    metas = [c["metadata"] for c in cfgs]
    if all(any("T:" in x for x in m) for m in metas):
        temps = [float(next(x for x in m if "T:" in x).split(":")[1]) for m in metas]
    in_models = None
    if all(any("model:" in x for x in m) for m in metas):
        in_models = [next(x for x in m if "model:" in x).split(":")[1] for m in metas]
    if args.get("density") is not None:                                            # IO.jl:255-260
        lam = [(rho / args["density"]) ** (1 / d) for rho in densities]
        densities = [args["density"]] * len(cfgs)
        positions = [p * l for p, l in zip(positions, lam)]
        boxes = [b * l for b, l in zip(boxes, lam)]
    if args.get("temperature") is not None:
        t = args["temperature"]
        temps = list(t) if isinstance(t, (list, tuple, np.ndarray)) else [t] * len(cfgs)
    elif temps is None:
        raise KeyError("temperature array has not been found in metadata or is not defined")
    if args.get("model") is not None:
        mm = args["model"]
        in_models = list(mm) if isinstance(mm, (list, tuple)) else [mm] * len(cfgs)
    elif in_models is None:
        raise KeyError("model array has not been found in metadata or is not defined")
    positions = [fold_back(p, b) for p, b in zip(positions, boxes)]                # IO.jl:284
    nsim = args.get("nsim") or 1
    if nsim > 1:                                                                   # IO.jl:287-294
        positions = [p.copy() for p in positions for _ in range(nsim)]
        species = [s.copy() for s in species for _ in range(nsim)]
        densities = [r for r in densities for _ in range(nsim)]
        temps = [t for t in temps for _ in range(nsim)]
    n_species = len(set(int(s) for sp in species for s in sp))
    model_matrix = parse_model(in_models[0], n_species)
    maxcut = max(m.rcut for row in model_matrix for m in row)
    Z = np.mean(densities) * _volume_sphere(maxcut, d)
    list_name = "LinkedList" if Z / N < 0.1 else "EmptyList"                       # IO.jl:309-310
    if args.get("list_type") is not None:
        list_name = args["list_type"]
    list_type = LIST_TYPES.get(list_name)
    if list_type is None:
        raise NotImplementedError(f"unknown list_type {list_name} (LinkedList, CellList, VerletList, EmptyList)")
    chains = [System(positions[k], species[k], densities[k], temps[k], model_matrix, list_type=list_type,
                     list_parameters=args.get("list_parameters"), seed=seed, chain=k, device=device)
              for k in range(len(positions))]
    return chains


# ---- writers (IO.jl:332-372, xyz.jl:79-84, exyz.jl:91-96, lammps.jl:88-106) -------------
def _jl_vec(v):
    return ",".join(repr(float(x)) for x in v)


def write_header(io, system, t, fmt, digits=6):
    if isinstance(fmt, XYZ):
        io.write(f"{len(system)}\n")
        io.write(f"step:{t} columns:species,position dt:1 cell:{_jl_vec(system.box)} rho:{system.density!r} "
                 f"T:{system.temperature!r}\n")
    elif isinstance(fmt, EXYZ):
        io.write(f"{len(system)}\n")
        b = [repr(float(x)) for x in system.box]
        if len(b) == 2:
            lat = f"{b[0]} 0.0 0.0 0.0 {b[1]} 0.0 0.0 0.0 0.0"
        else:
            lat = f"{b[0]} 0.0 0.0 0.0 {b[1]} 0.0 0.0 0.0 {b[2]}"
        io.write(f'Lattice="{lat}" Properties=:species:S:1:pos:R:{system.d} Time={t}\n')
    else:
        io.write("ITEM: TIMESTEP\n")
        io.write(f"{t}\n")
        io.write("ITEM: NUMBER OF ATOMS\n")
        io.write(f"{len(system)}\n")
        io.write("ITEM: BOX BOUNDS pp pp pp\n")
        for a in range(system.d):
            io.write(f"0.0 {float(system.box[a])!r}\n")
        if system.d == 2:
            io.write("-0.1 0.1\n")
            io.write("ITEM: ATOMS  type x y\n")
        else:
            io.write("ITEM: ATOMS  type x y z\n")


def store_trajectory(io, system, t, fmt, digits=6):
    """Arianna.store_trajectory(io, system::Atoms, t, format) -- src/IO/IO.jl:365-372."""
    write_header(io, system, t, fmt, digits)
    pos, sp = system.position, system.species
    for s, x in zip(sp, pos):
        io.write(f"{s}" + "".join(f" {v:.{digits}f}" for v in x) + "\n")


store_lastframe = store_trajectory


def write_system(io, system):
    """Arianna.write_system -- src/IO/IO.jl:17-25."""
    io.write(f"\tNumber of particles: {system.N}\n")
    io.write(f"\tDimensions: {system.d}\n")
    io.write(f"\tCell: {list(system.box)}\n")
    io.write(f"\tDensity: {system.density}\n")
    io.write(f"\tTemperature: {system.temperature}\n")
    io.write(f"\tNeighbour list: {type(system.neighbour_list).__name__}\n")
