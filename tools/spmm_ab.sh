cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in "" c4; do echo "== variant='$v'"; DN_LIB_VARIANT=$v timeout 120 python tools/microbench.py --reps 30 2>&1 | grep -E "grad_apply"; done; done
