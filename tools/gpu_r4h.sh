cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in tb_h tb_hg tb_te1 tb_te2 tb_te3; do echo -n "$v: "; timeout 60 tools/ubench/$v; done; done
