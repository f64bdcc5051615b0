cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "ring_gemm or geglu or linear_shapes" > gpurun_out/r3_t1.log 2>&1; echo "rc=$?"; tail -n 4 gpurun_out/r3_t1.log
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_l_$tag.json 2> gpurun_out/r3_bench_l_$tag.err; }
run ring X=1
run noring CRG_GEMM_RING=0
run ring2 X=1
run noring2 CRG_GEMM_RING=0
echo done
