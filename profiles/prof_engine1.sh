#!/bin/bash
# Round 4: rocprofv3 --stats of the literal one-wavefront loop (engine 1, k_ref_ops) -- the kernel VERDICT r03 "weak 2" names.
#   gpurun --timeout 600 -- 'bash profiles/prof_engine1.sh'   then   cp gpurun_out/seq_summary/r04_seq_engine1_* profiles/
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/prof; mkdir -p $P gpurun_out/seq_summary; rm -rf $P/e1_trace
cat > $P/e1_run.py <<'PY'
import json, sys, time
sys.path.insert(0, ".")
import particlesmc_amd as pmc
from particlesmc_amd.workloads import lj_lattice
pos, sp, rho = lj_lattice(32, jitter=0.05, seed=42)
s = pmc.System(pos, sp, rho, 1.0, pmc.LennardJones(1.0, 1.0), seed=42)
pool = [pmc.Move(pmc.Displacement(0, None, 0.0), pmc.SimpleGaussian(), {"sigma": 0.05}, 1.0)]
pmc.mc_sweep(s, pool, nsweeps=20, mode="checkerboard")
s.set_sequential_engine(1, 0)
t0 = time.perf_counter(); st = pmc.mc_sweep(s, pool, nsweeps=1, mode="sequential"); dt = time.perf_counter() - t0
print(json.dumps({"N": len(pos), "engine": 1, "steps": len(pos), "moves_per_s": len(pos) / dt, "us_per_step": 1e6 * dt / len(pos), "kernel_ms": st.kernel_ms}))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $P/e1_trace -- python3 $P/e1_run.py > gpurun_out/seq_summary/r04_seq_engine1_run.json 2> $P/e1.err
f=$(find $P/e1_trace -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/seq_summary/r04_seq_engine1_N32768_kernel_stats.csv
rm -rf $P/e1_trace
cat gpurun_out/seq_summary/r04_seq_engine1_run.json; head -4 gpurun_out/seq_summary/r04_seq_engine1_N32768_kernel_stats.csv
