"""The bench's N > 1 step on a real RCCL communicator. No second GPU is available to these tests, so the communicator has ONE rank and
the N > 1 branches of hcflow_amd/dist.py are driven by telling THAT module the world has two ranks: what runs for the first time on
hardware is then everything but the peer -- the asynchronous all_gather_into_tensor on RCCL's stream behind a call whose two half
batches ran on two side streams, its wait before the output buffer is reused, the barrier + MAX all-reduce of the timing contract on
device tensors, and the DDP training step of config 5 (tests/test_gpu_backward.py). The sharding arithmetic itself is covered with real
world-2 process groups on CPU (tests/test_dist_cpu.py, gloo)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


class _TwoRankView:
    """torch.distributed as hcflow_amd.dist sees it, with get_world_size() = 2 (everything else is the real module)."""

    def __init__(self, real):
        self._real = real

    def get_world_size(self, group=None):
        return 2

    def __getattr__(self, name):
        return getattr(self._real, name)


def test_bench_step_and_timing_contract_over_rccl(monkeypatch):
    import torch.distributed as dist
    import hcflow_amd.dist as hd
    from hcflow_amd import HCFlowNet_SR
    from hcflow_amd.config import preset
    from tests.util import cached_params
    cfg = preset("SR_4X_tiny")
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(cached_params("SR_4X_tiny", 11), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").eval()
    g = torch.Generator().manual_seed(3)
    B = 6                                                    # >= 4: the call runs as two half batches on the two side streams
    lr = torch.rand(B, 3, 24, 32, generator=g).cuda()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        with torch.no_grad():
            # the sharded call and the job-wide seed on the communicator as it is (one rank: its shard is the batch)
            assert hd.common_seed(lr.device, None, 1234) == 1234
            got = hd.sharded_inverse(net, lr, 0.8, seed=41)
            assert torch.equal(got, net(lr=lr, z=None, u=None, eps_std=0.8, reverse=True, seed=41, sample_offset=0))
        monkeypatch.setattr(hd, "dist", _TwoRankView(dist))
        out_all = torch.full((B, 3, 96, 128), -1.0, device="cuda:0")
        outs = {}

        def step(i):
            outs[i] = hd.gathered_step(net, lr, 0.8, 500 + i, out_all, overlap=True)

        with torch.no_grad():
            dt = hd.timed_region(step, 3, first=0)
            assert dt > 0 and not hd._pending               # the last step's gather was waited for inside the timed window
            torch.cuda.synchronize()
            assert torch.equal(out_all, outs[2])             # the gathered batch of a one-rank communicator = this rank's shard
            want = net(lr=lr, z=None, u=None, eps_std=0.8, reverse=True, seed=502, sample_offset=0)
            assert torch.equal(outs[2], want)
            assert not torch.equal(outs[1], outs[2])         # (another seed per step)
            # the blocking form completes out_all before it returns
            out_all.fill_(-1.0)
            o = hd.gathered_step(net, lr, 0.8, 777, out_all, overlap=False)
            torch.cuda.synchronize()
            assert torch.equal(out_all, o)
    finally:
        monkeypatch.undo()
        dist.destroy_process_group()
