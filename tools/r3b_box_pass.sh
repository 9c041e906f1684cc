#!/bin/bash
# one box's figures with the round's final kernels: the default bench line three times (three processes = three placements), c2, c3, mesh
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r3b_box_$(date +%H%M%S).jsonl; : > $out
for i in 1 2 3; do timeout -k 10 400 python bench.py --no-cpu-baseline >> $out 2>/dev/null; done
for w in c2 c3 mesh; do timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline >> $out 2>/dev/null; done
python - $out <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln); r = d['roofline']; c = r.get('cold_single_shot') or {}; sr = r.get('this_box_stream_read') or {}
    print(f"{d['config']['name']:6s} us {r['launch_us']:8.2f} frac {r['frac']:.4f} cold {c.get('launch_us')} {c.get('frac')} GF {d['value']:8.1f} box stream {sr.get('gbs')} bitwise {d.get('parity',{}).get('bitwise')}")
PY
