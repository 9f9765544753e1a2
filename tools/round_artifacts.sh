# Round artefacts on one box: kernel-trace statistics, HBM traffic and matrix-pipe counter passes (each its own run), for ViT-L
# (the headline) and ViT-S, the fp32 storage mode's kernel trace, plus the FETCH_SIZE calibration on a GEMM whose A operand is
# read exactly once.  The profiled runs keep the encoder's frame groups on one stream, as the bench's own roofline pass does.
# (The bench line itself, with its minutes of CPU baseline, is a call of its own: python bench.py > gpurun_out/<P>_bench.json.)
set -e
P=${1:-r3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-end-to-end --no-cpu-baseline --no-vits --no-fp32-mode --encoder-streams 1"
rm -rf gpurun_out/kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o res -- $B --steps 5 --warmup 2 > gpurun_out/${P}_vitl518_bench_under_rocprof.json 2> gpurun_out/kt.err
python tools/rocpd_stats.py $(ls gpurun_out/kt/*.db | tail -n 1) gpurun_out/${P}_vitl518_kernel_stats.csv
python tools/gpu_busy.py $(ls gpurun_out/kt/*.db | tail -n 1) gpurun_out/${P}_vitl518_gpu_busy.json
rm -rf gpurun_out/kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o res -- $B --encoder vits --steps 20 --warmup 3 > gpurun_out/${P}_vits518_bench_under_rocprof.json 2> gpurun_out/kt.err
python tools/rocpd_stats.py $(ls gpurun_out/kt/*.db | tail -n 1) gpurun_out/${P}_vits518_kernel_stats.csv
python tools/gpu_busy.py $(ls gpurun_out/kt/*.db | tail -n 1) gpurun_out/${P}_vits518_gpu_busy.json
rm -rf gpurun_out/kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o res -- python3 bench.py --precision fp32 --no-end-to-end --no-cpu-baseline --no-vits --encoder-streams 1 --steps 3 --warmup 1 > gpurun_out/${P}_vitl518_fp32_bench_under_rocprof.json 2> gpurun_out/kt.err
python tools/rocpd_stats.py $(ls gpurun_out/kt/*.db | tail -n 1) gpurun_out/${P}_vitl518_fp32_kernel_stats.csv
rm -rf gpurun_out/kt gpurun_out/pf gpurun_out/pw
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pf -o res -- $B --steps 1 --warmup 1 --no-kernel-timing > gpurun_out/pmc_f.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pw -o res -- $B --steps 1 --warmup 1 --no-kernel-timing > gpurun_out/pmc_w.log 2>&1
python tools/pmc_traffic.py $(ls gpurun_out/pf/*.db | tail -n 1) $(ls gpurun_out/pw/*.db | tail -n 1) gpurun_out/${P}_vitl518_hbm_traffic.json
rm -rf gpurun_out/pf gpurun_out/pw gpurun_out/pm
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pm -o res -- $B --steps 1 --warmup 1 --no-kernel-timing > gpurun_out/pmc_m.log 2>&1
python tools/pmc_mfma.py $(ls gpurun_out/pm/*.db | tail -n 1) gpurun_out/${P}_vitl518_mfma_util.json
rm -rf gpurun_out/pm gpurun_out/pc
# calibration: M = 43840, K = 1024, N = 256 on the 256x256 kernel: one column tile, so A (89.8 MB) is fetched exactly once through
# global_load_lds; W (0.5 MB) stays in L2
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pc -o res -- python3 tools/gemm_one.py 2 256 1024 43840 3 > gpurun_out/pmc_c.log 2>&1
python tools/pmc_traffic.py $(ls gpurun_out/pc/*.db | tail -n 1) $(ls gpurun_out/pc/*.db | tail -n 1) gpurun_out/${P}_fetch_calibration_glds.json
rm -rf gpurun_out/pc
