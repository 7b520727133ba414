#!/bin/bash
# Rounds 4-5: profiles of the reference-order chain, the team instance, the big-neighbourhood and single-cell instances (PMC_SEQUENTIAL; pmc_seq.h).  Run through gpurun from the repo root:
#   gpurun --timeout 1800 -- 'bash profiles/collect_seq.sh'
# It summarises ON the box (the raw traces exceed what gpurun merges back) into gpurun_out/seq_summary/; then, here:
#   cp gpurun_out/seq_summary/* profiles/
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
# rounds of the reference-order windows launch by launch under the profiler (hipGraphLaunch + --kernel-trace crashes in the tool
# for some windows on ROCm 7.2); the library itself never looks for a profiler
export PMC_SEQ_NOGRAPH=1
P=gpurun_out/prof
mkdir -p $P
# ---- round 4: the reference-order chain (PMC_SEQUENTIAL; pmc_seq.h) ----------------------------------------------------------
# engine 2 (speculative windows) on the headline system, engine 3 (state in LDS) on 1000 chains of the reference's benchmark system
rm -rf $P/seq_small_trace $P/seq_trace $P/seq_many_trace $P/seq_sq $P/seq_many_sq $P/seq_fetch $P/seq_write
export STEPS=3000000 NO_ENGINE1=1
rocprofv3 --kernel-trace --stats --output-format csv -d $P/seq_trace -- python3 examples/seq_rate.py 100 > $P/seq_rate.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $P/seq_many_trace -- python3 examples/seq_many.py 6 1000 100 > $P/seq_many.log 2>&1
# round 5: a small window, one launch per round (k_sq_round_fused; N = 4096, windows of 64)
rm -rf $P/seq_small_trace
STEPS=1000000 rocprofv3 --kernel-trace --stats --output-format csv -d $P/seq_small_trace -- python3 examples/seq_rate.py 16 > $P/seq_small.log 2>&1
export STEPS=1000000
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $P/seq_sq -- python3 examples/seq_rate.py 100 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/seq_fetch -- python3 examples/seq_rate.py 100 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/seq_write -- python3 examples/seq_rate.py 100 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $P/seq_many_sq -- python3 examples/seq_many.py 6 1000 20 > /dev/null 2>&1
STEPS=2000000 python3 examples/seq_rate.py 16 32 48 64 100 > gpurun_out/seq_rate_table.jsonl 2>&1
for n in 1 64 256 1000 4000; do python3 examples/seq_many.py 6 $n 100; done > gpurun_out/seq_many_table.jsonl 2>&1
python3 examples/seq_many.py 10 256 20 >> gpurun_out/seq_many_table.jsonl 2>&1
# the team-of-8 instance of the throughput kernel (config 2), counters for DESIGN 8b / VERDICT r03 item 4
rm -rf $P/teams_sq $P/teams_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $P/teams_trace -- python3 profiles/c2_run.py > $P/teams_run.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $P/teams_sq -- python3 profiles/c2_run.py > /dev/null 2>&1
# round 5: the big-neighbourhood instance (published mixture at N = 8000) and the single-cell box (the same example as published, N = 1000)
for w in big single; do
  rm -rf $P/${w}_sq $P/${w}_trace
  a=""; [ $w = single ] && a=single
  rocprofv3 --kernel-trace --stats --output-format csv -d $P/${w}_trace -- python3 profiles/mix_run.py $a > $P/${w}_run.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $P/${w}_sq -- python3 profiles/mix_run.py $a > /dev/null 2>&1
done
# round 6: engine 4 (VerletList chains in LDS, k_sq_verlet) on the reference's CI system: one chain of 100 sweeps, 256 chains of 20
rm -rf $P/verlet_trace $P/verlet_many_trace $P/verlet_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $P/verlet_trace -- python3 examples/verlet_rate.py 100 1 > $P/verlet.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $P/verlet_many_trace -- python3 examples/verlet_rate.py 20 256 > $P/verlet_many.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $P/verlet_sq -- python3 examples/verlet_rate.py 100 1 > /dev/null 2>&1
python3 examples/verlet_rate.py 100 1 16 64 256 1024 > gpurun_out/verlet_rate_table.jsonl 2>&1
SWAPS=1 python3 examples/verlet_rate.py 100 1 256 >> gpurun_out/verlet_rate_table.jsonl 2>&1
# round 6: engine 5 (small EmptyList chains in LDS, k_sq_empty) on the reference's published benchmark system: one chain, 256 chains
rm -rf $P/empty_trace $P/empty_many_trace $P/empty_sq
ENGINES=5 rocprofv3 --kernel-trace --stats --output-format csv -d $P/empty_trace -- python3 examples/empty_rate.py 1000 1 > $P/empty.log 2>&1
ENGINES=5 rocprofv3 --kernel-trace --stats --output-format csv -d $P/empty_many_trace -- python3 examples/empty_rate.py 200 256 > $P/empty_many.log 2>&1
ENGINES=5 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $P/empty_sq -- python3 examples/empty_rate.py 1000 1 > /dev/null 2>&1
python3 examples/empty_rate.py 1000 1 16 256 1000 > gpurun_out/empty_rate_table.jsonl 2>&1
mkdir -p gpurun_out/seq_summary
PMC_PROFILES_OUT=gpurun_out/seq_summary python3 profiles/summarize.py ${ROUND:-r06} --seq-only
rm -rf $P/seq_small_trace $P/seq_trace $P/seq_many_trace $P/seq_sq $P/seq_many_sq $P/seq_fetch $P/seq_write $P/teams_sq $P/teams_trace $P/big_sq $P/big_trace $P/single_sq $P/single_trace $P/verlet_trace $P/verlet_many_trace $P/verlet_sq $P/empty_trace $P/empty_many_trace $P/empty_sq
ls gpurun_out/seq_summary
