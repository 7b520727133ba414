"""Parallel tempering across GPUs: one replica per rank, RCCL only for the exchange step.

The reference has no replica exchange ("parallel tempering (soon)", README.md:33), so this
defines it: every `exchange_every` sweeps each rank all-gathers ONE double (its
total energy) -- `torch.distributed.all_gather`, i.e. RCCL over xGMI with the nccl backend,
gloo on CPU -- then every rank evaluates the same Philox-keyed Metropolis decisions for the
active (even/odd) pairs of adjacent temperatures with `pmc_pt_decide` and adopts its new
temperature label.  Configurations never move between GPUs; the bulk traffic is zero.
Acceptance: min(1, exp((beta_a - beta_b) (E_a - E_b))).
"""
import ctypes as C

import numpy as np

from . import _lib


def pt_decide(seed, rnd, energies, ladder, label):
    """Deterministic exchange decisions (same answer on every rank). Returns (label, n_accepted)."""
    e = np.ascontiguousarray(energies, dtype=np.float64)
    lad = np.ascontiguousarray(ladder, dtype=np.float64)
    lab = np.ascontiguousarray(label, dtype=np.int32).copy()
    nacc = C.c_int32()
    _lib.check(_lib.lib().pmc_pt_decide(int(seed), int(rnd), len(e), e.ctypes.data_as(C.POINTER(C.c_double)),
                                        lad.ctypes.data_as(C.POINTER(C.c_double)),
                                        lab.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nacc)))
    return lab, nacc.value


class ReplicaExchange:
    """State of one rank in a temperature ladder of `world` replicas."""

    def __init__(self, ladder, seed, rank=0, world=1, group=None, device="cpu"):
        self.ladder = np.ascontiguousarray(ladder, dtype=np.float64)
        if len(self.ladder) != world:
            raise ValueError("one temperature per rank")
        self.seed, self.rank, self.world, self.group, self.device = int(seed), rank, world, group, device
        self.label = np.arange(world, dtype=np.int32)     # replica r starts at ladder[r]
        self.round = 0
        self.accepted = 0
        self.attempted = 0

    @property
    def temperature(self):
        return float(self.ladder[self.label[self.rank]])

    def gather(self, energy):
        if self.world == 1:
            return np.array([energy], dtype=np.float64)
        import torch
        import torch.distributed as dist
        mine = torch.tensor([energy], dtype=torch.float64, device=self.device)
        out = [torch.zeros(1, dtype=torch.float64, device=self.device) for _ in range(self.world)]
        dist.all_gather(out, mine, group=self.group)
        return np.array([float(t.item()) for t in out], dtype=np.float64)

    def exchange(self, energy):
        """One exchange round; returns this rank's new temperature."""
        energies = self.gather(float(energy))
        self.label, nacc = pt_decide(self.seed, self.round, energies, self.ladder, self.label)
        parity = self.round & 1
        self.attempted += max(0, (self.world - parity) // 2)
        self.accepted += nacc
        self.round += 1
        return self.temperature


def _env_int(names, default):
    import os
    for n in names:
        if n in os.environ:
            return int(os.environ[n])
    return default


def _launch_key():
    """What every rank of ONE launch shares and no other launch has: the launcher's run / job id."""
    import os
    e = os.environ
    if e.get("TORCHELASTIC_RUN_ID"):
        return "te" + e["TORCHELASTIC_RUN_ID"] + "_" + e.get("MASTER_PORT", "")
    for k in ("PMIX_NAMESPACE", "OMPI_MCA_ess_base_jobid", "OMPI_MCA_orte_ess_jobid"):
        if e.get(k):
            return "ompi" + e[k]
    if e.get("SLURM_JOB_ID"):
        return "slurm" + e["SLURM_JOB_ID"] + "." + e.get("SLURM_STEP_ID", "0")
    if e.get("MASTER_PORT"):
        return "port" + e["MASTER_PORT"]
    return None


def rccl_comm_from_env(device, tag="pmc", id_file=None):
    """One process per GPU without torch or MPI bindings: rank 0 creates the RCCL unique id (pmc_rccl_unique_id) and hands its
    128 bytes to the other ranks of the node through a file; every rank then joins through pmc_rccl_comm_create.

    Rank and world size are read from RANK / WORLD_SIZE (torchrun), OMPI_COMM_WORLD_RANK / _SIZE (mpirun) or SLURM_PROCID /
    SLURM_NTASKS (srun).  The file is `id_file`, else $PMC_RCCL_ID_FILE, else a name in $TMPDIR built from the launch key (the
    launcher's run / job id: TORCHELASTIC_RUN_ID, PMIX_NAMESPACE, SLURM_JOB_ID.SLURM_STEP_ID, MASTER_PORT) and, for launchers
    whose ranks share a parent, that parent's PID; a world of more than one rank with NO launch key must name the file
    (nothing would tell this launch's id from a crashed or concurrent job's).  The file carries, after the 128 bytes, the launch
    key and rank 0's start time: a rank only takes a file whose key is its own and whose stamp is not older than its own start
    by more than two minutes.  Rank 0 removes any file of that name BEFORE it creates the id and removes its own file once
    ncclCommInitRank has returned (it returns only after every rank has joined, i.e. has read the id)."""
    import os
    import struct
    import time
    t_start = time.time()
    rank = _env_int(("RANK", "OMPI_COMM_WORLD_RANK", "SLURM_PROCID"), 0)
    world = _env_int(("WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "SLURM_NTASKS"), 1)
    path = id_file or os.environ.get("PMC_RCCL_ID_FILE")
    key = _launch_key() or ""
    if not path:
        if world > 1 and not key:
            raise RuntimeError("rccl_comm_from_env: more than one rank and no launcher run id in the environment: name the id "
                               "file (PMC_RCCL_ID_FILE or id_file=), one per launch")
        shared_parent = "RANK" in os.environ and "OMPI_COMM_WORLD_RANK" not in os.environ and "SLURM_PROCID" not in os.environ
        owner = os.getpid() if world == 1 else (os.getppid() if shared_parent else 0)
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"{tag}_rccl_id_{key or 'single'}_{owner}")
    kbytes = key.encode()[:64].ljust(64, b"\0")
    if rank == 0:
        try:
            os.remove(path)                      # a stale id of an earlier job with this name
        except FileNotFoundError:
            pass
        uid = RcclComm.unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid + kbytes + struct.pack("<d", t_start))
        os.replace(path + ".tmp", path)
    else:
        t0 = time.time()
        while True:
            try:
                blob = open(path, "rb").read()
                if len(blob) == 128 + 64 + 8 and blob[128:192] == kbytes and struct.unpack("<d", blob[192:])[0] > t_start - 120.0:
                    uid = blob[:128]
                    break
            except FileNotFoundError:
                pass
            if time.time() - t0 > 120:
                raise TimeoutError(f"no RCCL id of this launch from rank 0 at {path}")
            time.sleep(0.05)
    comm = RcclComm(uid, rank, world, device)    # collective: returns once every rank has joined
    if rank == 0:
        try:
            os.remove(path)
        except FileNotFoundError:
            pass
    return comm


def run_tempering(system, pool, ladder, nsweeps, exchange_every, seed, rank=0, world=1, group=None, device="cpu",
                  mode="checkerboard", comm=None):
    """Sweeps on this rank's replica with an exchange round every `exchange_every` sweeps.  With `comm` (an RcclComm)
    the round is pmc_replica_exchange_rccl -- the library's ncclAllGather, no torch; otherwise torch.distributed
    gathers the energies (RCCL with the nccl backend, gloo on CPU) and pmc_pt_decide decides."""
    from .simulation import mc_sweep
    rex = ReplicaExchange(ladder, seed, rank, world, group, device)
    system.temperature = rex.temperature
    done = 0
    trace = []
    while done < nsweeps:
        k = min(exchange_every, nsweeps - done)
        st = mc_sweep(system, pool, nsweeps=k, mode=mode)
        done += k
        if comm is not None:
            rex.label, nacc = comm.exchange(system, rex.ladder, rex.seed, rex.round, rex.label)   # sets the handle's temperature
            rex.attempted += max(0, (world - (rex.round & 1)) // 2)
            rex.accepted += nacc
            rex.round += 1
        else:
            system.temperature = rex.exchange(st.energy)
        trace.append((done, int(rex.label[rank]), st.energy))
    return rex, trace


def replica_exchange(systems, ladder, seed, rnd, label):
    """pmc_replica_exchange: one round over replicas held as handles of this process.  The library reads the
    handles' running energies, decides and sets their temperatures.  Returns (label, n_accepted)."""
    n = len(systems)
    hs = (C.c_void_p * n)(*[s._h for s in systems])
    lad = np.ascontiguousarray(ladder, dtype=np.float64)
    lab = np.ascontiguousarray(label, dtype=np.int32).copy()
    nacc = C.c_int32()
    _lib.check(_lib.lib().pmc_replica_exchange(hs, n, lad.ctypes.data_as(C.POINTER(C.c_double)), int(seed), int(rnd),
                                               lab.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nacc)))
    return lab, nacc.value


class RcclComm:
    """An RCCL communicator created through the library's own entry points (pmc_rccl_*), for hosts without torch."""

    def __init__(self, unique_id, rank, world, device):
        self.rank, self.world = rank, world
        self._c = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        _lib.check(_lib.lib().pmc_rccl_comm_create(buf, rank, world, device, C.byref(self._c)))

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        _lib.check(_lib.lib().pmc_rccl_unique_id(buf))
        return bytes(buf)

    def exchange(self, system, ladder, seed, rnd, label):
        """pmc_replica_exchange_rccl: ncclAllGather of one double per rank, identical decisions on every rank."""
        lad = np.ascontiguousarray(ladder, dtype=np.float64)
        lab = np.ascontiguousarray(label, dtype=np.int32).copy()
        nacc = C.c_int32()
        _lib.check(_lib.lib().pmc_replica_exchange_rccl(system._h, self._c, self.rank, self.world,
                                                        lad.ctypes.data_as(C.POINTER(C.c_double)), int(seed), int(rnd),
                                                        lab.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nacc)), system._h)
        return lab, nacc.value

    def close(self):
        if self._c:
            _lib.check(_lib.lib().pmc_rccl_comm_destroy(self._c))
            self._c = C.c_void_p()


def run_tempering_local(systems, pool, ladder, nsweeps, exchange_every, seed, mode="checkerboard"):
    """Parallel tempering with all replicas as handles of this process (one GPU, or several
    devices through the handles' `device`): same decisions as the one-rank-per-GPU driver, the
    energies simply need no collective (pmc_replica_exchange).  Returns (labels, accepted, attempted,
    per-replica traces of (temperature index, energy))."""
    from .simulation import mc_sweep_many
    n = len(systems)
    ladder = np.ascontiguousarray(ladder, dtype=np.float64)
    label = np.arange(n, dtype=np.int32)
    for r, s in enumerate(systems):
        s.temperature = float(ladder[label[r]])
    done, rnd, acc, att = 0, 0, 0, 0
    traces = [[] for _ in range(n)]
    while done < nsweeps:
        k = min(exchange_every, nsweeps - done)
        for r, st in enumerate(mc_sweep_many(systems, pool, nsweeps=k, mode=mode)):
            traces[r].append((int(label[r]), st.energy))
        done += k
        label, nacc = replica_exchange(systems, ladder, seed, rnd, label)
        att += max(0, (n - (rnd & 1)) // 2)
        acc += nacc
        rnd += 1
    return label, acc, att, traces
