set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for pf in 0 1 0 1; do SPR_ATTN_PF=$pf NSEG=64 SPR_ATTN_MODES=1,3 timeout -k 10 100 python scripts/attn_abl.py 2>&1 | grep mode | sed "s/^/PF=$pf /"; done | tee gpurun_out/attn_pf.log
SPR_ATTN_PF=1 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_range.py -x -q -m gpu -k "attention or attn" 2>&1 | tail -3
