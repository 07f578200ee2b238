#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
C4="--preset Rescaling_DF2K_4X --batch 8 --lr-size 160 --steps 10 --warmup 3 --no-other-precision --no-cpu-baseline --no-other-configs"
for rep in 1 2; do
python bench.py $C4 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fat16   ', j['value'], j['ms_per_step'], j['precision']['check']['max_abs_diff_f16x3_vs_exact_f32'], [(k['kernel'][:34],k['launches_per_step'],k['ms_per_step']) for k in j['roofline']['conv_kernels'][:4]])"
HCF_NO_FAT=1 python bench.py $C4 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('per-conv', j['value'], j['ms_per_step'], j['precision']['check']['max_abs_diff_f16x3_vs_exact_f32'], [(k['kernel'][:34],k['launches_per_step'],k['ms_per_step']) for k in j['roofline']['conv_kernels'][:4]])"
done
python -m pytest tests/test_gpu_nets.py tests/test_gpu_wino.py tests/test_gpu_f16x3.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_real.py tests/test_gpu_engine.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tail -8
