#!/bin/bash
# A/B/C... of prebuilt library variants on ONE GPU box (box-to-box clock variance is +-3 %).
# usage (via gpurun): tools/gpu_ab_multi.sh TAG "lib1.so lib2.so ..." [bench args]
#   pass 1: every variant with a short CPU parity leg (parity_ok must be true), 8-bit and 10-bit
#   pass 2: every variant again without it (timing only)
TAG=$1; LIBS=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cp rav1e_amd/librav1e_hip.so /tmp/lib_orig.so
run() { # lib bd cpu_seconds
  cp $1 rav1e_amd/librav1e_hip.so
  python bench.py --steps 20 --warmup 3 --no-extra --cpu-seconds $3 --bit-depth $2 "${@:4}" 2>&1 | tail -1 > /tmp/ab_line.json
  python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open('/tmp/ab_line.json').read())
    print("%-24s bd %2s  %9.0f Mpx/s  kernel_ms %s  parity %s" % (sys.argv[1].split('/')[-1], sys.argv[2], d['value'], d['kernel_ms'], d.get('parity_ok')))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", open('/tmp/ab_line.json').read()[-400:])
PY
}
for bd in 8 10; do
  for v in $LIBS; do run $v $bd 2 "$@"; done
done 2>&1 | tee $OUT/pass1.txt
for rep in 1 2; do
  for bd in 8 10; do
    for v in $LIBS; do run $v $bd 0 "$@"; done
  done
done 2>&1 | tee $OUT/pass2.txt
cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so
