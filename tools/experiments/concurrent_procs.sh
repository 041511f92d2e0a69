mkdir -p gpurun_out/conc
run2() { # tag B
  for i in 1 2; do VC_BENCH_B=$2 python bench.py --steps $3 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/conc/$1_$i.json 2> gpurun_out/conc/$1_$i.err & done; wait
}
VC_BENCH_B=128 python bench.py --steps 600 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/conc/solo128.json 2>/dev/null
run2 two128 128 1200
run2 two64 64 2400
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/conc/*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['value']), round(d['ms_per_step'],3))
    except Exception as e: print(f, 'ERR', e)
PY
