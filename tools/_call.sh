cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_models.py -x -q -m gpu -k "fp16_operand" -s > gpurun_out/r3_t2.log 2>&1; echo "rc=$?"; grep parity gpurun_out/r3_t2.log; tail -n 2 gpurun_out/r3_t2.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_k_bf16.json 2> gpurun_out/r3_bench_k_bf16.err; echo "rc=$?"
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --half f16 > gpurun_out/r3_bench_k_f16.json 2> gpurun_out/r3_bench_k_f16.err; echo "rc=$?"
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --unet-fp32 > gpurun_out/r3_bench_k_fp32.json 2> gpurun_out/r3_bench_k_fp32.err; echo "rc=$?"
