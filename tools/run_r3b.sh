set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "pieces or saturation or layernorm or split3 or fp32_mode or three_term or bilinear" > gpurun_out/r3b_kernels.log 2>&1
timeout -k 10 500 python -u -m pytest tests/test_model_gpu.py -x -q --timeout 300 --timeout-method thread -k "fp32 or clamped or outliers" > gpurun_out/r3b_model.log 2>&1
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-vits --no-kernel-timing > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err
bash tools/prof_fp32.sh r3b
