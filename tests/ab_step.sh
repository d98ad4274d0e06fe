# whole-step A/B of the small-layer choices: tile threshold (VAMD_NT_WIDE_MIN) x split reduction (VAMD_SPLITK_MAX_TILES)
for P in flowers ffhq; do for W in 1; do for S in 104 136 200; do
echo "== $P VAMD_NT_WIDE_MIN=$W VAMD_SPLITK_MAX_TILES=$S"
VAMD_NT_WIDE_MIN=$W VAMD_SPLITK_MAX_TILES=$S timeout -k 10 200 python bench.py --preset $P --no-cpu-baseline --steps ${STEPS:-10} --warmup 3 --sample-steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('sampling',{}).get('image_steps_per_s'))" || exit 1
done; done; done
