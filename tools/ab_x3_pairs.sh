C="--steps 12 --warmup 4 --no-cpu-baseline --no-forward-only --no-trainer-window --no-live-pmc --no-extras --no-roofline --n-sp 10000 --n-edges 50000 --n-feat 11 --model-config gru_10,f_8 --precision bf16x3"
for t in "" "14:1" "14:1,17:1" ; do
  for i in 1 2; do python bench.py $C ${t:+--tune $t} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tune=$t', d['ms_per_step'], d['value'])"; done
done
