# A/B helper: same box, alternating runs.  usage: [REPS=3 STEPS=60] bash tools/experiments/ab.sh "VAR=a" "VAR=b" ...
for rep in $(seq 1 ${REPS:-2}); do for v in "$@"; do
  echo -n "$v: "; env $v timeout 300 python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['stage_ms_per_step'])"
done; done
