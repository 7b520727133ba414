#!/bin/bash
# Collects the per-round evidence on an MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash profiles/collect.sh'
# then, back in the build container:  python profiles/summarize.py rNN
# --pmc passes are separate runs and never share a command line with a trace domain other than --kernel-trace.
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof; rm -rf gpurun_out/prof/trace gpurun_out/prof/fetch gpurun_out/prof/write gpurun_out/prof/sq
python3 bench.py --steps 20 --warmup 5 > gpurun_out/prof/bench_full.json 2> gpurun_out/prof/bench_full.err
# the stats summary of this run averages over EVERY k_cb_pass launch (lattice start, 1000 equilibration sweeps, warm-up, timed
# steps); summarize.py also averages the per-dispatch trace over the last 20 launches only -- the timed region of bench.py
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/prof/trace_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/fetch -- python3 bench.py --steps 4 --warmup 1 --equilibrate 600 --no-cpu > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/write -- python3 bench.py --steps 4 --warmup 1 --equilibrate 600 --no-cpu > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/prof/sq -- python3 bench.py --steps 4 --warmup 1 --equilibrate 600 --no-cpu > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof/sq2 -- python3 bench.py --steps 4 --warmup 1 --equilibrate 600 --no-cpu > /dev/null 2>&1
python3 bench.py --steps 100 --warmup 5 --equilibrate 2000 --no-cpu > gpurun_out/prof/bench_equilibrated.json 2> /dev/null
python3 bench.py --steps 20 --warmup 5 --equilibrate 0 --no-cpu > gpurun_out/prof/bench_lattice.json 2> /dev/null
python3 examples/throughput_table.py > gpurun_out/throughput_table.jsonl 2>&1
ls gpurun_out/prof
