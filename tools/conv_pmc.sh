# PMC passes (separate runs, --kernel-trace only) of one conv shape under the ring (CRG_RING=2) and the staggered (6) schedule.
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in 2 6; do
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  CRG_RING=$mode timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_conv_${mode}_$tag -o run -- python3 $R/tools/conv_one.py 640 640 64 > $R/gpurun_out/pmc_conv_${mode}_$tag.log 2>&1
done
done
echo done
