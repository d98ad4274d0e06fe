# same-box A/B of two builds of the library: bench.py alternately with $1 and $2 (paths of libvision_amd.so), $3 rounds
A=$1; B=$2; N=${3:-2}
for i in $(seq $N); do
  for L in $A $B; do
    VAMD_LIB=$L timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 |
      python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$L'.rsplit('/', 1)[-1], d['value'], 'img/s', d['roofline']['achieved'], 'TF NT', (d.get('sampling') or {}).get('image_steps_per_s'))" || exit 1
  done
done
