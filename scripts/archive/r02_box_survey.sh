#!/bin/bash
# one line per box: what the memory system gives for the headline launch's bytes, and the compose kernel at A1 / A2
O=gpurun_out/boxsurvey; mkdir -p $O
tag=$(date +%H%M%S)
./scripts/ubench/stream_pattern 4096 > $O/stream_$tag.txt 2>&1
q() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f %.3f' % (d['roofline']['kernel_ms_mean'], d['roofline']['frac']))"; }
a1=$(python bench.py --regime A1 --cpu-sample 8 --pristine-launches 0 2>/dev/null | q)
a2=$(python bench.py --regime A2 --cpu-sample 8 --pristine-launches 0 2>/dev/null | q)
copy=$(grep "^copy " $O/stream_$tag.txt | tail -1 | awk '{print $2, $6}' | tr -d '(')
pat=$(grep "^pattern" $O/stream_$tag.txt | tail -1 | awk '{print $2, $6}' | tr -d '(')
mer=$(grep "^merged" $O/stream_$tag.txt | tail -1 | awk '{print $2, $6}' | tr -d '(')
echo "box $tag | copy $copy | real addresses $pat | merged $mer | compose A1 $a1 | compose A2 $a2" | tee -a $O/summary.txt
