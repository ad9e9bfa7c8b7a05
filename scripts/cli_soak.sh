#!/bin/bash
# leak check of the batch pipeline: peak RSS must not grow with the number of files (GPU box)
cd "$(dirname "$0")/.."
work=$(mktemp -d /tmp/mjq.XXXX); mkdir -p $work/in $work/out
H=libmodjpeg_amd/host
T=tests/tools
for i in $(seq 1 16); do BENCH_DUMP_INPUT=$work/in/img$i.jpg MJ_AMD=no-gpu-decode,no-gpu-encode $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 16 16 1 > /dev/null; done
cp tests/golden/images/dropon.png $work/logo.png
for n in ${SIZES:-8192 65536}; do
  rm -f $work/list.txt; for i in $(seq 1 $n); do echo $work/in/img$(( (i % 16) + 1 )).jpg >> $work/list.txt; done
  python3 - "$H/modjpeg" "$work" <<'PY'
import resource, subprocess, sys
exe, work = sys.argv[1], sys.argv[2]
r = subprocess.run([exe, "--batch", work + "/list.txt", "--outdir", work + "/out", "--jobs", "16", "-p", "br", "-m", "-10,-10", "-d", work + "/logo.png", "-y", "10", "-O"],
                   capture_output=True, text=True)
print(r.stderr.strip().splitlines()[-1])
print("peak RSS of the process: %.0f MB" % (resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1024.0))
PY
done
rm -rf $work
