cd ${GRAFT_REPO_ROOT:-$(pwd)}
for b in 1 8; do HIFICAR_PROFILE_DETAIL=1 python tools/batch_profile.py --batch $b 2>/dev/null; done
