line() { python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-8s %-22s %-3s img %-5s step %.4f ms  frac(plan bytes) %.3f  parity %s | %s' % ('$1', d['config']['workload'], d['config']['alpha_regime'], d['config']['images_total'], d['ms_per_step'], r['frac'], d['parity'], d['config']['launch'][:60]))"; }
for w in "compose_4k_batch A5" "compose_4k_batch A4" "compose_1080p_batch A5 --images 1024" "compose_1080p_batch A5 --images 512" "compose_1080p_batch A4 --images 512" "compose_1080p_batch A5 --images 256"; do set -- $w; wl=$1; rg=$2; shift 2
  for launch in serial auto; do python bench.py --workload $wl --regime $rg "$@" --launch $launch --steps 30 --cpu-sample 1 --pristine-launches 0 2>/dev/null | line $launch; done
done
