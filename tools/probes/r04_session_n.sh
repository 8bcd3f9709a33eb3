cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
python3 tools/hier_bench.py 256 20 > gpurun_out/r04_n_hier256.txt 2>&1; cat gpurun_out/r04_n_hier256.txt
cd /tmp && export TMPDIR=/tmp
SUHMO_GRAPH_MAX_CELLS=0 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04_n_prof -o h -- python3 $R/tools/hier_bench.py 256 10 > $R/gpurun_out/r04_n_prof.log 2>&1
cd $R
T=$(ls gpurun_out/r04_n_prof/*kernel_trace.csv gpurun_out/r04_n_prof/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/trace_busy.py $T 0.4 > gpurun_out/r04_n_busy.txt 2>&1
python3 tools/stats_by_grid.py $T 13 > gpurun_out/r04_n_by_grid.txt 2>&1
head -30 gpurun_out/r04_n_busy.txt
python3 - "$T" <<'PY'
import csv,sys
rows=sorted(csv.DictReader(open(sys.argv[1])),key=lambda r:int(r["Start_Timestamp"]))
n=len(rows); seg=rows[int(n*0.8):int(n*0.8)+400]
t0=int(seg[0]["Start_Timestamp"]); pe=None
out=open("gpurun_out/r04_n_sequence.txt","w")
for r in seg:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    nm=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][:46]
    g=int(r["Grid_Size_X"])*int(r["Grid_Size_Y"])*int(r["Grid_Size_Z"])
    out.write("%-46s grid %8d start %9.1f dur %6.1f gap %6.1f\n"%(nm,g,(s-t0)/1e3,(e-s)/1e3,(s-pe)/1e3 if pe else 0)); pe=e
PY
rm -rf gpurun_out/r04_n_prof
