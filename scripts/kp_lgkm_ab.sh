# A/B of the ring kernel's slot-refill ordering wait (s_waitcnt lgkmcnt(0) in front of the refill DMAs):
#   make -C superpoints_registration_amd/csrc OUT=../libspr_hip_nolgkm.so BUILD=build/nolgkm/ EXTRA=-DSPR_KP_RING_NO_LGKM
# then, on the GPU box:
cd $GRAFT_REPO_ROOT
for lib in libspr_hip.so libspr_hip_nolgkm.so; do
  echo "== $lib"
  SPR_HIP_LIB=$PWD/superpoints_registration_amd/$lib PAIRS=32 LEVELS=0,1 timeout -k 10 300 python scripts/kpconv_ring_bench.py 2>&1 | grep '^L'
done
