set -o pipefail
cd $GRAFT_REPO_ROOT
for b in 32 64 96 128 64 32; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-train-leg --no-extra-legs --pairs-per-step $b 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('pairs/step', d['config']['pairs_per_step_per_gpu'], 'value', d['value'], 'ms/step', d['ms_per_step'], 'kp frac', d['roofline']['frac'], 'attn', d['roofline_attention']['frac'])"; done
