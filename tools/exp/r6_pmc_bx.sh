cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc_bx
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVES"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_bx/$n -o pmc -- python3 $R/tools/bench_kernels.py --what bf16x3 > $R/gpurun_out/pmc_bx/$n.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(R+"/gpurun_out/pmc_bx/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bf16x3" not in r["Kernel_Name"]: continue
        key=(r["Kernel_Name"][:60], r["Grid_Size"])
        agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        if r["Counter_Name"]=="GRBM_GUI_ACTIVE": cnt[key]+=1
for k,v in agg.items():
    n=cnt[k] or 1
    print(k, n, {a: round(b/n) for a,b in sorted(v.items())})
PY
