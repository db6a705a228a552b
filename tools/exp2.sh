cd $GRAFT_REPO_ROOT
for i in 1 2; do tools/ubench/store_pattern 4096; echo; done
