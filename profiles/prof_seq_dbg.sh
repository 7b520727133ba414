cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python -m pytest tests/test_gpu_sequential.py -x -q 2>&1 | tail -2
for d in 0; do
P=gpurun_out/prof_seq_$d; rm -rf $P; mkdir -p $P
PMC_SEQ_DBG=$d STEPS=1000000 NO_ENGINE1=1 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 examples/seq_rate.py 100 > $P/run.log 2>&1
echo "dbg=$d"; find $P/trace -name "*kernel_stats.csv" | head -1 | xargs grep -E "k_sq" | cut -d, -f1-4
done
