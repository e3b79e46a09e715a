cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6j
for n in 64 128 192 256 384; do
  for narrow in 1 0; do
    rm -rf gpurun_out/r6j/kt
    RS_GEMM_B3J_NARROW=$narrow RS_GEMM_B3J_MR=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6j/kt -- python profiles/micro/tdnnf_decode.py 3 1 $n > /dev/null 2>&1
    f=$(find gpurun_out/r6j/kt -name "*kernel_trace.csv" | head -1)
    python - <<PY
import csv
rows=sorted(csv.DictReader(open("$f")), key=lambda r:int(r['Start_Timestamp']))
seq=[(r['Kernel_Name'][53:72], round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1), int(r.get('Grid_Size_X') or r.get('Grid_Size'))//256) for r in rows if 'GemmKernelB3' in r['Kernel_Name']]
print("utts $n narrow $narrow:", seq[-15:-11])
PY
  done
done
