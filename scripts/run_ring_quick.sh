# KPConv parity tests + ring-vs-r2 timings
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_range.py -x -q -k kpconv > gpurun_out/ring_t.log 2>&1; echo test_rc=$?; tail -3 gpurun_out/ring_t.log
timeout -k 10 300 python scripts/kpconv_ring_bench.py > gpurun_out/ring_b.log 2>&1; echo bench_rc=$?; grep '^L' gpurun_out/ring_b.log
