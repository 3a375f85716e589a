# ablation builds of the ring kernel (ab/libspr_*.so, see SPR_KP_ABL_* in kpconv.hip), conv launches of levels 0 and 1
cd $GRAFT_REPO_ROOT
for v in "" NODMA NOMFMA1 NOP2 NODMA_NOMFMA1; do
  if [ -z "$v" ]; then lib=$PWD/superpoints_registration_amd/libspr_hip.so; else lib=$PWD/ab/libspr_$v.so; fi
  echo "== variant ${v:-full}"
  SPR_HIP_LIB=$lib LEVELS=0,1 KINDS=neighbors timeout -k 10 200 python scripts/kpconv_ring_bench.py 2>&1 | grep '^L' | cut -c1-120
done

