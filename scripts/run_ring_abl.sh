# experiment builds of the ring kernel (ab/libspr_*.so), conv launches of levels 0 and 1
cd $GRAFT_REPO_ROOT
for v in "" ${VARIANTS:-NS5 PRIO NS3} ""; do
  if [ -z "$v" ]; then lib=$PWD/superpoints_registration_amd/libspr_hip.so; else lib=$PWD/ab/libspr_$v.so; fi
  echo "== variant ${v:-default}"
  SPR_HIP_LIB=$lib LEVELS=0,1 KINDS=neighbors timeout -k 10 200 python scripts/kpconv_ring_bench.py 2>&1 | grep '^L' | cut -c1-120
done
