for spw in 3 1; do for split in 1 2; do
python3 bench.py --no-cpu-baseline --no-secondary --steps 400 --warmup 100 --streams-per-gpu 8192 --split $split --kernel $spw 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('S 8192 kernel $spw split $split: step_us %.2f (min %.2f) frac %.3f' % (1000*d['ms_per_step'], 1000*d['timing']['ms_per_step_min'], d['roofline']['frac']))"
done; done
