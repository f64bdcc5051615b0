set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in 8 5; do
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  CRG_ATTN_DMA=$mode timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_attn_${mode}_$tag -o run -- python3 $R/tools/attn_one.py 8 > $R/gpurun_out/pmc_attn_${mode}_$tag.log 2>&1
done
done
echo done
