cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "transposed_range" > gpurun_out/r3_t1.log 2>&1; echo "rc=$?"; tail -n 3 gpurun_out/r3_t1.log
timeout -k 10 200 python tools/qkv_vt_probe.py > gpurun_out/r3_qkvvt.txt 2>&1; grep -v amdgpu gpurun_out/r3_qkvvt.txt
timeout -k 10 900 python -m pytest tests/test_hip_models.py -x -q -m gpu -k "transformer or unet_small or unet_sd15_full or sgm_unet or cross_attention" > gpurun_out/r3_t2.log 2>&1; echo "rc=$?"; tail -n 3 gpurun_out/r3_t2.log
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_m_$tag.json 2> gpurun_out/r3_bench_m_$tag.err; }
run vt X=1
run novt CRG_SELF_VT=0
run vt2 X=1
run novt2 CRG_SELF_VT=0
