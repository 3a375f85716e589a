# one gpurun call: KPConv parity tests, ring-vs-r2 timings, phase stamps, rocprofv3 kernel stats
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_range.py -x -q -k kpconv > gpurun_out/ring_t.log 2>&1; echo test_rc=$?; tail -3 gpurun_out/ring_t.log
timeout -k 10 300 python scripts/kpconv_ring_bench.py > gpurun_out/ring_b.log 2>&1; echo bench_rc=$?; grep '^L' gpurun_out/ring_b.log
SPR_HIP_LIB=$PWD/ab/libspr_prof.so timeout -k 10 300 python scripts/kp_ring_prof.py > gpurun_out/ring_prof.log 2>&1; echo prof_rc=$?
grep -v amdgpu.ids gpurun_out/ring_prof.log | tail -22
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/ring_rp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ring_rp -o ring -- python3 $GRAFT_REPO_ROOT/scripts/kpconv_ring_bench.py > $GRAFT_REPO_ROOT/gpurun_out/ring_rp.log 2>&1; echo rp_rc=$?
cd $GRAFT_REPO_ROOT; python scripts/kstats.py $(find gpurun_out/ring_rp -name '*kernel_stats.csv' | head -1) 1 12 2>&1 | tail -14
