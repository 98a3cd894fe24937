cd $GRAFT_REPO_ROOT
for pf in 1 thread; do echo "PREFETCH=$pf"; PREFETCH=$pf python tools/diag_train.py 60 2>/dev/null | tail -5; done
