cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CRG_GEMM_D_MAX=384 CRG_ROWRES=0 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_d0.json 2> gpurun_out/r3_bench_d0.err
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_d1.json 2> gpurun_out/r3_bench_d1.err
CRG_GEMM_D_MAX=384 CRG_ROWRES=0 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_d0b.json 2> gpurun_out/r3_bench_d0b.err
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_d1b.json 2> gpurun_out/r3_bench_d1b.err
echo done
