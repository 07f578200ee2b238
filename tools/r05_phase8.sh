#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_p8; mkdir -p $O
{
python tools/two_stream_probe.py --steps 8 --ways 2 2>/dev/null
python tools/two_stream_probe.py --steps 8 --ways 2 --threads 0 2>/dev/null
python tools/two_stream_probe.py --steps 8 --ways 4 2>/dev/null
python tools/two_stream_probe.py --steps 20 --ways 2 --preset SR_CelebA_8X --batch 32 --lr-size 20 2>/dev/null
python tools/two_stream_probe.py --steps 20 --ways 4 --preset SR_CelebA_8X --batch 32 --lr-size 20 2>/dev/null
python tools/two_stream_probe.py --steps 20 --ways 2 --preset SR_CelebA_8X --batch 32 --lr-size 20 --threads 0 2>/dev/null
} | tee $O/two_stream.txt
