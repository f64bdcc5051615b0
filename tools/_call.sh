cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --workload sdxl --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r3_w_sdxl_prof.json 2> gpurun_out/r3_w_sdxl_prof.err; echo rc=$?
