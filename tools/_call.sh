cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3_ff.txt
: > $O
timeout -k 10 200 python tools/ff_probe.py >> $O 2>&1
CRG_LIB=tools/ab/libcrg_nogelu.so timeout -k 10 200 python tools/ff_probe.py >> $O 2>&1
timeout -k 10 200 python tools/ff_probe.py >> $O 2>&1
CRG_LIB=tools/ab/libcrg_nogelu.so timeout -k 10 200 python tools/ff_probe.py >> $O 2>&1
echo done
