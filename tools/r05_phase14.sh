#!/bin/bash
# round 5, phase 14: the second half batch enqueued by a helper thread; GPU_MAX_HW_QUEUES=8 against the queue oversubscription
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p14
mkdir -p $O
cd $GRAFT_REPO_ROOT
COMMON="--no-other-precision --no-cpu-baseline --no-exact-check"
run() { tag=$1; shift; env "$@" python bench.py --steps 8 --warmup 3 $COMMON --other-steps 10 > $O/b_$tag.json 2> $O/b_$tag.err; }
run s1 HCFLOW_STREAMS=1
run s2 HCFLOW_STREAMS=2
run s2q8 HCFLOW_STREAMS=2 GPU_MAX_HW_QUEUES=8
run s1q8 HCFLOW_STREAMS=1 GPU_MAX_HW_QUEUES=8
python - <<PY
import json
for t in ("s1","s2","s2q8","s1q8"):
    try:
        j=json.loads(open("$O/b_%s.json"%t).read().strip().splitlines()[-1])
    except Exception as e:
        print(t,"FAILED",e); continue
    oc=j.get("other_configs",{})
    print(t,j["value"],j["ms_per_step"],"| c1",oc.get("config1_single_patch_latency",{}).get("value"),"c3",oc.get("config3_face_x8_tau_sweep",{}).get("value"),
          "c4",oc.get("config4_rescaling_roundtrip",{}).get("value"),"c5",oc.get("config5_nll_train_step",{}).get("ms_per_step"),oc.get("config5_nll_train_step",{}).get("phases_ms"), oc.get("error"))
PY
python -m pytest tests/test_gpu_engine.py -m gpu -q -x -k "two_stream" 2>&1 | tail -2
