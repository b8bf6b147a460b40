for i in 1 2; do for V in "" "--tune 14:1"; do
python bench.py --precision bf16x3 --n-sp 10000 --n-edges 50000 --n-feat 11 --model-config gru_10,f_8 --steps 10 --warmup 3 --no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline --no-extras $V 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$V]', round(d['ms_per_step'],4), round(d['value']))"
done; done
