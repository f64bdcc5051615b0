set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "gn_stats" > gpurun_out/r3_t1.log 2>&1; echo "ops rc=$?"; tail -n 3 gpurun_out/r3_t1.log
timeout -k 10 900 python -m pytest tests/test_hip_models.py -x -q -m gpu -k "resblock or updown or transformer or unet_small or unet_sd15_full or controlnet or graph or c4_unit" > gpurun_out/r3_t2.log 2>&1; echo "models rc=$?"; tail -n 3 gpurun_out/r3_t2.log
CRG_GN_STATS=0 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r3_bench_gn0.json 2> gpurun_out/r3_bench_gn0.err; echo "bench0 rc=$?"
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r3_bench_gn1.json 2> gpurun_out/r3_bench_gn1.err; echo "bench1 rc=$?"
