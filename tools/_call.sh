cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python tools/lin_probe.py > gpurun_out/r3_lin3.txt 2>&1
timeout -k 10 200 python tools/c1_probe.py > gpurun_out/r3_c1b.txt 2>&1
CRG_GN_STATS=0 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_f0.json 2> gpurun_out/r3_bench_f0.err
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_f1.json 2> gpurun_out/r3_bench_f1.err
CRG_GN_STATS=0 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_f0b.json 2> gpurun_out/r3_bench_f0b.err
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_f1b.json 2> gpurun_out/r3_bench_f1b.err
timeout -k 10 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "linear or conv or gn_stats" > gpurun_out/r3_t1.log 2>&1; echo "ops rc=$?"; tail -n 3 gpurun_out/r3_t1.log
