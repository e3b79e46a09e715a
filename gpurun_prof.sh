cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d gpurun_out/prof_sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_write.log 2>&1
find gpurun_out -name "*.csv" | head -30
