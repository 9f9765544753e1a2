set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/s4_gpu_tests.log 2>&1
tail -n 2 gpurun_out/s4_gpu_tests.log
timeout -k 10 400 python bench.py > gpurun_out/s4_bench.json 2> gpurun_out/s4_bench.err
tail -n 1 gpurun_out/s4_bench.json | cut -c1-400
rm -rf gpurun_out/s4_kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/s4_kt -o res -- python3 bench.py --steps 5 --warmup 2 --no-end-to-end > gpurun_out/s4_bench_under_rocprof.json 2> gpurun_out/s4_kt.err
python tools/rocpd_stats.py $(ls gpurun_out/s4_kt/*.db | tail -n 1) gpurun_out/s4_kernel_stats.csv
rm -rf gpurun_out/s4_kt gpurun_out/s4_f gpurun_out/s4_w
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/s4_f -o res -- python3 bench.py --steps 1 --warmup 1 --no-end-to-end > gpurun_out/s4_pmc_f.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/s4_w -o res -- python3 bench.py --steps 1 --warmup 1 --no-end-to-end > gpurun_out/s4_pmc_w.log 2>&1
python tools/pmc_traffic.py $(ls gpurun_out/s4_f/*.db | tail -n 1) $(ls gpurun_out/s4_w/*.db | tail -n 1) gpurun_out/s4_hbm_traffic.json
rm -rf gpurun_out/s4_f gpurun_out/s4_w
rm -rf gpurun_out/s4_m
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/s4_m -o res -- python3 bench.py --steps 1 --warmup 1 --no-end-to-end > gpurun_out/s4_pmc_m.log 2>&1
python tools/pmc_mfma.py $(ls gpurun_out/s4_m/*.db | tail -n 1) gpurun_out/s4_mfma_util.json
rm -rf gpurun_out/s4_m
