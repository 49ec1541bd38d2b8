cd $GRAFT_REPO_ROOT
python -m pytest tests/test_box_fused_gpu.py -x -q -m gpu 2>&1 | tail -2
for cfg in "EFG_BOX_DETERMINISTIC=1" "EFG_BOX_DETERMINISTIC=0" "EFG_BOX_DETERMINISTIC=1 EFG_BOX_OFF_SCALE=3"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg bash scripts/box_time.sh $tag | grep -E "encoder|tile|bin"
done
