for E in X=1 TVM_DOT_GX=1 TVM_DOT_G=4; do
  env $E python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$E', d['ms_per_step'], d['stage_ms']['out-of-domain rows'], d['stage_ms']['linear combination'])"
done
