cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in tb_g16 tb_g8 tb_g4 tb14_g16 tb14_g8 tb14_g4; do echo -n "$v: "; timeout 60 tools/ubench/$v; done; done
