# A/B helper: same box, alternating runs.  usage: bash tools/ab.sh "VAR=a" "VAR=b" ...
for rep in 1 2; do for v in "$@"; do
  echo -n "$v: "; env $v timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['stage_ms_per_step'])"
done; done
