cd $GRAFT_REPO_ROOT
cat /proc/loadavg
export EFG_DETERMINISTIC=1
for r in 1 2; do
for leg in "EFG_GEOM_PREFETCH=0" "EFG_GEOM_PREFETCH=1" "EFG_VOX_OWN_STATE=0" "EFG_CONV_SMALL=0"; do
  echo "$leg: $(env $leg timeout 200 python scripts/ubench/soak.py 150 2>&1 | tail -1 | cut -c1-110)"
done
done
cat /proc/loadavg
