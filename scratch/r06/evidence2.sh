# final-kernel evidence: timelines first (short), then evidence.sh
cd $GRAFT_REPO_ROOT
bash scratch/r06/final_tl.sh
bash scratch/r06/evidence.sh
