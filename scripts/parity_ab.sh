for e in "EFG_X=0" "EFG_FUSED_ENCODER=0" "EFG_FUSED_ENCODER=0 EFG_BOX_SHARED_PROJ=0"; do
  env $e python bench.py --steps 2 --warmup 1 --no-full-graph --no-arm 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['parity_full_size']; print('$e', p['max_rel_diff'], p['worst_term'], p['grad_norm_rel_diff'], p['total_cpu'], p['total_gpu'])"
done
