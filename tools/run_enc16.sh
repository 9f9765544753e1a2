# fp16 against bf16 encoder operands in the 16-bit mode: kernel tests of the attention, the 720p margin, the model suites, speed
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k attention --timeout 120 --timeout-method thread > gpurun_out/enc16_attn.log 2>&1
for dt in bf16 fp16; do
  VDA_ENCODER_DTYPE=$dt timeout -k 10 300 python -u tools/margin_720p.py half > gpurun_out/enc16_margin_$dt.log 2>&1
  VDA_ENCODER_DTYPE=$dt timeout -k 10 300 python bench.py --no-end-to-end --no-cpu-baseline --no-fp32-mode --steps 10 --warmup 3 > gpurun_out/enc16_bench_$dt.json 2> gpurun_out/enc16_bench_$dt.err
done
VDA_ENCODER_DTYPE=fp16 timeout -k 10 900 python -u -m pytest tests/test_model_gpu.py tests/test_modules_gpu.py -q -m gpu --timeout 300 --timeout-method thread -rA > gpurun_out/enc16_model.log 2>&1 || true
