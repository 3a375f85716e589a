set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for a in 0 1 2 4 3 7; do SPR_HIP_LIB=$GRAFT_REPO_ROOT/superpoints_registration_amd/libspr_hip_abl.so SPR_ATTN_ABL=$a NSEG=64 SPR_ATTN_MODES=1 timeout -k 10 100 python scripts/attn_abl.py 2>&1 | grep mode | sed "s/^/ABL=$a /"; done | tee gpurun_out/attn_ablate.log
for rep in 1 2; do
echo "== new"; PAIRS=16 LEVELS=0,1 timeout -k 10 300 python scripts/kpconv_ring_bench.py 2>&1 | grep "^L"
echo "== old"; SPR_HIP_LIB=$GRAFT_REPO_ROOT/superpoints_registration_amd/libspr_hip_kpold.so PAIRS=16 LEVELS=0,1 timeout -k 10 300 python scripts/kpconv_ring_bench.py 2>&1 | grep "^L"
done | tee gpurun_out/kpring_ab.log
