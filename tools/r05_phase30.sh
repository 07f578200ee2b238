#!/bin/bash
# with the two-stream default: is the fat pair (conv1 + conv2's old-input part) worth it at 320^2 now? (HCF_FAT12_PIXELS)
O=gpurun_out/r05_p30
mkdir -p $O
for rep in 1 2; do
for fp in 40000 110000; do
HCF_FAT12_PIXELS=$fp python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-precision --no-exact-check --no-other-configs > $O/fat_$fp.json 2> $O/fat_$fp.err
python - <<PY
import json
j=json.loads(open("$O/fat_$fp.json").read().strip().splitlines()[-1])
print("FAT12_PIXELS $fp:", j["value"], j["ms_per_step"], "single", j["single_stream"]["value"], [ (v["kernel"][:34], v["ms_per_step"]) for v in j["roofline"]["conv_kernels"][:4]])
PY
done
done
