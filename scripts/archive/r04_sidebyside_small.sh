line() { python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-13s %-3s img %-5s step %.4f ms  parity %s | %s' % ('$1', d['config']['alpha_regime'], d['config']['images_total'], d['ms_per_step'], d['parity'], d['config']['launch'][:50]))"; }
for rg in A5 A4; do for n in 512 256 128 64; do
  for launch in serial side-by-side; do python bench.py --regime $rg --images $n --launch $launch --steps 50 --cpu-sample 1 --pristine-launches 0 2>/dev/null | line $launch; done
done; done
