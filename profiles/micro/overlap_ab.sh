# which of the round-2 kernels cost throughput when four calls overlap (all-pdfs model, as round 1 measured it)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02ab
B="python bench.py --no-cpu-baseline --steps 300 --all-pdfs"
$B > gpurun_out/r02ab/all_on.json 2>/dev/null
RS_GEMM_B3I=0 $B > gpurun_out/r02ab/no_b3i.json 2>/dev/null
RS_UBM_MFMA=0 $B > gpurun_out/r02ab/no_ubm.json 2>/dev/null
RS_IVEC_MFMA=0 $B > gpurun_out/r02ab/no_f64.json 2>/dev/null
RS_TRIM_HALO=0 $B > gpurun_out/r02ab/no_trim.json 2>/dev/null
RS_GEMM_B3I=0 RS_UBM_MFMA=0 RS_IVEC_MFMA=0 RS_TRIM_HALO=0 $B > gpurun_out/r02ab/all_off.json 2>/dev/null
