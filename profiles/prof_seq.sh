cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/prof_seq
rm -rf $P; mkdir -p $P
STEPS=2000000 NO_ENGINE1=1 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 examples/seq_rate.py 100 > $P/run.log 2>&1
find $P/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $P/kernel_stats.csv
head -20 $P/kernel_stats.csv
cat $P/run.log | tail -3
# per-kernel gap analysis from the trace: mean start-to-start of consecutive kernels
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_seq/trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if 'k_sq' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
import collections
gap=collections.defaultdict(list); dur=collections.defaultdict(list)
for a,b in zip(rows,rows[1:]):
    ka=a['Kernel_Name'].split('(')[0][:24]; kb=b['Kernel_Name'].split('(')[0][:24]
    gap[(ka,kb)].append(int(b['Start_Timestamp'])-int(a['End_Timestamp']))
    dur[ka].append(int(a['End_Timestamp'])-int(a['Start_Timestamp']))
for k,v in dur.items(): print('dur',k,len(v),sum(v)/len(v)/1e3,'us')
for k,v in gap.items(): print('gap',k,len(v),sum(v)/len(v)/1e3,'us')
PY
