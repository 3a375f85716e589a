cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_preprocess.py tests/test_gpu_config_shapes.py -x -q -m gpu > gpurun_out/t_knn.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/t_knn.log
for rep in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-train-leg --no-extra-legs --config kitti --generator lidar --pairs-per-step 8 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('kitti', d['value'], d['ms_per_step'])"; done
for rep in 1 2; do SPR_NBR_ALGO=wave timeout -k 10 300 python bench.py --no-cpu-baseline --no-train-leg --no-extra-legs 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('headline forced wave', d['value'], d['ms_per_step'], 'attn', d['roofline_attention']['frac'])"; done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-train-leg --no-extra-legs 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('headline default', d['value'], d['ms_per_step'], 'attn', d['roofline_attention']['frac'])"
