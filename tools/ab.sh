# usage: ab.sh "ENV1" "ENV2" ...   (each a space-separated env assignment list; "-" = none)
for e in "$@"; do
  if [ "$e" = "-" ]; then e=""; fi
  env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-aux > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(sys.argv[1] or '-', d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], d['roofline']['achieved'])" "$e" || tail -3 gpurun_out/ab.err
done
