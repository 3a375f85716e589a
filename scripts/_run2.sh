cd $GRAFT_REPO_ROOT
bash scripts/measure_round.sh 2>&1 | tail -2 | cut -c1-300
