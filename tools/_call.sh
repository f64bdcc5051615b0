set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_ctxp1 -o run -- python3 $R/tools/ctx_attn_probe.py > $R/gpurun_out/r3_ctx_probe_1.txt 2>&1 &&
CRG_ATTN_CTX=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_ctxp0 -o run -- python3 $R/tools/ctx_attn_probe.py > $R/gpurun_out/r3_ctx_probe_0.txt 2>&1
echo rc=$?
