# FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) of the persistent GEGLU GEMM under both tile orders, then device times.
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sup in 0 1; do
for grp in FETCH_SIZE WRITE_SIZE; do
  CRG_GEMM_RING_SUPER=$sup timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ring_${sup}_$grp -o run -- python3 $R/tools/geglu_one.py 8192 5120 640 > $R/gpurun_out/pmc_ring_${sup}_$grp.log 2>&1
  python3 $R/tools/pmc_sum.py $(find $R/gpurun_out/pmc_ring_${sup}_$grp -name "*counter_collection.csv" | head -1) gemm_ring | tee -a $R/gpurun_out/pmc_ring_summary.txt
done
done
cd $R
for shape in "8192 5120 640" "16384 5120 640" "4096 10240 1280" "32768 2560 320"; do
  for sup in 0 1 0 1; do CRG_GEMM_RING_SUPER=$sup python3 tools/geglu_one.py $shape time | tee -a gpurun_out/pmc_ring_summary.txt; done
done
