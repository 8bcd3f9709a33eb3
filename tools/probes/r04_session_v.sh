cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SUHMO_GRAPH_MAX_CELLS=0 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04_v_prof -o h -- python3 $R/tools/probes/amr3_via_hier.py > $R/gpurun_out/r04_v_prof.log 2>&1
cd $R
T=$(ls gpurun_out/r04_v_prof/*kernel_trace.csv gpurun_out/r04_v_prof/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/trace_busy.py $T 0.3 > gpurun_out/r04_v_busy.txt 2>&1
python3 - "$T" <<'PY'
import csv,sys,collections
rows=sorted(csv.DictReader(open(sys.argv[1])),key=lambda r:int(r["Start_Timestamp"]))
n=len(rows); seg=rows[int(n*0.75):]          # the hier variant's timed steps
agg=collections.OrderedDict()
for r in seg:
    nm=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][:46]
    g=int(r["Grid_Size_X"])*int(r["Grid_Size_Y"])*int(r["Grid_Size_Z"])
    a=agg.setdefault((nm,g),[0,0.0]); a[0]+=1; a[1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
tot=sum(v[1] for v in agg.values())
out=open("gpurun_out/r04_v_by_grid.txt","w")
out.write("# last quarter of the trace (hier entry points): %d kernels, %.1f ms of kernel time\n"%(len(seg),tot/1e3))
for (nm,g),(c,us) in sorted(agg.items(),key=lambda kv:-kv[1][1])[:45]:
    out.write("%-46s %9d %6d %8.1f %8.2f %5.1f%%\n"%(nm,g,c,us/c,us/1e3,100*us/tot))
PY
head -3 gpurun_out/r04_v_busy.txt
rm -rf gpurun_out/r04_v_prof
