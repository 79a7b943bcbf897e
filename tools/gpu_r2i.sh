R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15
LASR_FE_LEGACY=1 timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
