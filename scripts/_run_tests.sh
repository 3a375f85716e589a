set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/t_all.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/t_all.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
