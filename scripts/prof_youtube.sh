R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/py
GOCTR_NO_GRAPH=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/py/fetch -- python $R/bench.py --workload youtube --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>$R/gpurun_out/py/err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/py/kt -- python $R/bench.py --workload youtube --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>>$R/gpurun_out/py/err
python - <<'PY'
import csv,glob,collections,os
R=os.environ["GRAFT_REPO_ROOT"]
f=glob.glob(f"{R}/gpurun_out/py/fetch/*/*_counter_collection.csv")[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "rocclr" in r["Kernel_Name"]: continue
    acc[(r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k,v in acc.items(): print("FETCH_SIZE KiB", k, len(v), round(sum(v)/len(v),1))
f=glob.glob(f"{R}/gpurun_out/py/kt/*/*_kernel_stats.csv")[0]
print(open(f).read()[:1500])
PY
