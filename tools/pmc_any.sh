# usage: bash tools/pmc_any.sh <outdir-under-gpurun_out> <kernel-substring> <python-script + args...>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; FILT=$2; shift 2
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS" ; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p$i -- python3 $R/$@ > $OUT.p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "$FILT" not in k: continue
        key = k[:44] + " grid=" + row.get("Grid_Size", "?")
        agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
