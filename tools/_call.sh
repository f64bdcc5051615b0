cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03a
