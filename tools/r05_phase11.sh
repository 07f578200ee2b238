#!/bin/bash
# round 5, phase 9: two half batches on two streams as the module default (HCFLOW_STREAMS=1: off): same-box A/B, then the -m gpu suite
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p11
mkdir -p $O
cd $GRAFT_REPO_ROOT
COMMON="--no-other-precision --no-cpu-baseline --no-exact-check"
for rep in 1 2; do
  for v in 2 1; do
    HCFLOW_STREAMS=$v python bench.py --steps 10 --warmup 3 $COMMON --other-steps 6 > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err
  done
done
python - <<PY
import json
for rep in (1,2):
    for v in (2,1):
        try:
            j=json.loads(open("$O/b_%d_%d.json"%(v,rep)).read().strip().splitlines()[-1])
        except Exception as e:
            print("streams",v,"rep",rep,"FAILED",e); continue
        k=[x for x in j["roofline"]["conv_kernels"] if "wino4_kernel<0|1|2>" in x["kernel"]][0]
        oc=j.get("other_configs",{})
        print("streams",v,"rep",rep,j["value"],j["ms_per_step"],"wino4",k["launches_per_step"],k["ms_per_step"],k["avg_launch_us"],k["frac_of_yardstick"],
              "| c1",oc.get("config1_single_patch_latency",{}).get("value"),"c3",oc.get("config3_face_x8_tau_sweep",{}).get("value"),
              "c4",oc.get("config4_rescaling_roundtrip",{}).get("value"),"c5",oc.get("config5_nll_train_step",{}).get("ms_per_step"), oc.get("error"))
PY
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_nets.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -v "^shapes" | tail -15 > $O/pytest.log
tail -6 $O/pytest.log
