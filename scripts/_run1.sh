set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_range.py tests/test_gpu_xenc.py tests/test_gpu_regtr.py -x -q -m gpu -k "attention or attn or transformer or xenc or stack or layer or matches_reference or encoder" > gpurun_out/t_attn.log 2>&1; echo "pytest rc $?" ; tail -5 gpurun_out/t_attn.log
(NSEG=64 timeout -k 10 120 python scripts/attn_abl.py; NSEG=64 SPR_ATTN_CORE=h3 SPR_ATTN_MODES=1,2 timeout -k 10 120 python scripts/attn_abl.py; NSEG=64 timeout -k 10 120 python scripts/attn_abl.py) > gpurun_out/attn_abl.log 2>&1; cat gpurun_out/attn_abl.log | grep mode
