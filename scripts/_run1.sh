set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_block_tail.py tests/test_gpu_regtr.py tests/test_gpu_preprocess.py -x -q -m gpu -k "block_tail or encoder or matches_reference or side_stream or neighbors or neighbours or radius or pyramid or truncated or tie or coincident" > gpurun_out/t_tail.log 2>&1; echo "pytest rc $?" ; tail -5 gpurun_out/t_tail.log
for v in 0 1 0 1; do SPR_NO_NORM_FOLD=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-train-leg --no-extra-legs > gpurun_out/bench_f$v.json 2> gpurun_out/bench_f$v.err; python - $v <<'PY'
import json,sys
m=sys.argv[1]
d=json.loads(open(f'gpurun_out/bench_f{m}.json').read().strip().splitlines()[-1])
print('SPR_NO_NORM_FOLD', m, 'bench value', d['value'], 'ms/step', d['ms_per_step'])
PY
done
for v in 0 1 0 1; do SPR_NBR_LDS_SORT=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-train-leg --no-extra-legs --config kitti --generator lidar --pairs-per-step 8 > gpurun_out/bench_k$v.json 2> gpurun_out/bench_k$v.err; python - $v <<'PY'
import json,sys
m=sys.argv[1]
d=json.loads(open(f'gpurun_out/bench_k{m}.json').read().strip().splitlines()[-1])
print('KITTI SPR_NBR_LDS_SORT', m, 'bench value', d['value'], 'ms/step', d['ms_per_step'])
PY
done
