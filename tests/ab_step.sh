# whole-step A/B of one environment switch: VAR="NAME" VALUES="a b c" PRESETS="flowers ffhq" bash tests/ab_step.sh   ("-" = unset)
for P in ${PRESETS:-flowers ffhq}; do for V in $VALUES; do
echo "== $P $VAR=$V"
( [ "$V" = "-" ] || export $VAR=$V; timeout -k 10 200 python bench.py --preset $P --no-cpu-baseline --steps ${STEPS:-10} --warmup 3 --sample-steps 10 2>/dev/null ) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('sampling',{}).get('image_steps_per_s'))" || exit 1
done; done
