R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/sq
GOCTR_NO_GRAPH=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/sq/a -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>$R/gpurun_out/sq/err
GOCTR_NO_GRAPH=1 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/sq/b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>>$R/gpurun_out/sq/err
python - <<'PY'
import csv,glob,collections,os
R=os.environ["GRAFT_REPO_ROOT"]
for leg in "ab":
    f=glob.glob(f"{R}/gpurun_out/sq/{leg}/*/*_counter_collection.csv")
    if not f: print("no file", leg); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=(r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        if "rocclr" in k[0]: continue
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
