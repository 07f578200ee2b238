"""Host-side checks of hcflow_amd.optim that need no GPU: argument validation as torch.optim.Adam's, and the loud failure on
CPU tensors (there is no CPU path)."""
import pytest
import torch

from hcflow_amd import _lib, optim


def test_adam_refuses_cpu_parameters():
    with pytest.raises(_lib.HcfError, match="no CPU path"):
        optim.Adam([torch.nn.Parameter(torch.zeros(3))], lr=1e-3)


@pytest.mark.parametrize("kw", [dict(lr=-1.0), dict(eps=-1e-8), dict(betas=(1.0, 0.9)), dict(betas=(0.9, -0.1)), dict(weight_decay=-1.0),
                                dict(amsgrad=True)])
def test_adam_argument_checks(kw):
    with pytest.raises(ValueError):
        optim.Adam([torch.nn.Parameter(torch.zeros(3))], **kw)


def test_clip_refuses_cpu_gradients():
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    with pytest.raises(_lib.HcfError, match="GPU only"):
        optim.clip_grad_norm_([p], 1.0)
    with pytest.raises(_lib.HcfError, match="GPU only"):
        optim.clip_grad_value_([p], 1.0)


def test_chunk_record_matches_the_header():
    import re, os
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "hcflow.h")).read()
    assert int(re.search(r"#define HCF_ADAM_CHUNK (\d+)", hdr).group(1)) == optim._CHUNK
    assert optim._chunk_dtype.itemsize == 16 and optim._chunk_dtype.fields["offset"][1] == 8 and optim._chunk_dtype.fields["n"][1] == 12
