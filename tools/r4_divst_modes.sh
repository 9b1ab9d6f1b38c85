#!/bin/bash
# divST sub-block time of the B=32 forward with and without the precise-CLS side path, one box
export TMPDIR=/tmp
O=gpurun_out/r4k; mkdir -p $O
for c in 1 0; do
  timeout 300 python bench.py --workload visual_fwd --steps 20 --warmup 5 --no-cpu-baseline --no-parity --cls-precise $c > $O/visual_cls$c.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/visual_cls$c.json'));r=d['roofline']['divst_subblock'];print('cls-precise $c:', d['ms_per_step'],'ms', d['value'],'clips/s | divST', r['ms'], 'ms', r['frac'], '| end-to-end', r['ms_end_to_end'], r['frac_end_to_end'])"
done
