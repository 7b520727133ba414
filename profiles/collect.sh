#!/bin/bash
# Collects the per-round evidence on an MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash profiles/collect.sh'
# then, back in the build container:  python profiles/summarize.py rNN
# --pmc passes are separate runs and never share a command line with a trace domain other than --kernel-trace.
# `--only-headline` makes the LAST steps / rounds launches of the sweep kernel the timed region of the run.
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
# rounds of the reference-order windows launch by launch under the profiler (hipGraphLaunch + --kernel-trace crashes in the tool
# for some windows on ROCm 7.2); the library itself never looks for a profiler
export PMC_SEQ_NOGRAPH=1
P=gpurun_out/prof
mkdir -p $P; rm -rf $P/trace $P/trace_m1 $P/fetch $P/write $P/sq $P/sq2 $P/sq_m1
python3 bench.py > $P/bench_full.json 2> $P/bench_full.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py --steps 200 --warmup 20 --only-headline --no-cpu > $P/trace_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_m1 -- python3 bench.py --steps 200 --warmup 20 --rounds 1 --only-headline --no-cpu > $P/trace_m1_bench.log 2>&1
PM="python3 bench.py --steps 16 --warmup 4 --equilibrate 600 --only-headline --no-cpu"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch -- $PM > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write -- $PM > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $P/sq -- $PM > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/sq2 -- $PM > /dev/null 2>&1
# the default schedule of the library (m = 1: what rounds 1-2 profiled), 4 timed launches
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $P/sq_m1 -- python3 bench.py --steps 4 --warmup 1 --rounds 1 --equilibrate 600 --only-headline --no-cpu > /dev/null 2>&1
python3 bench.py --steps 1000 --warmup 20 --equilibrate 2000 --long-steps 0 --no-cpu > $P/bench_equilibrated.json 2> /dev/null
python3 bench.py --steps 20 --warmup 5 --rounds 1 --equilibrate 0 --long-steps 0 --no-cpu > $P/bench_lattice.json 2> /dev/null
python3 examples/throughput_table.py > gpurun_out/throughput_table.jsonl 2>&1
PMC_ROUNDS=4 python3 examples/throughput_table.py > gpurun_out/throughput_table_m4.jsonl 2>&1
python3 tests/measure_move_efficiency.py > $P/move_efficiency.log 2>&1
ls $P
