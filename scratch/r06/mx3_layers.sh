cd $GRAFT_REPO_ROOT
for rep in 1 2; do for l in "" "30" "1,2" "1" "1,2,30"; do MX3_LAYERS=$l python scratch/mx3_layers.py 2>&1 | grep "MX3 on"; done; done
