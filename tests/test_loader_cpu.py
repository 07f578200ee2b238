"""hcflow_amd/loader.py: the batch > 1 replacement of the reference's batch-1 test loader (codes/data/__init__.py:24)."""
import torch

from hcflow_amd.loader import batched_test_loader


class FakeSet:
    """Items shaped like the reference's GTLQ datasets return them (data/GTLQ_dataset.py: LQ, GT, LQ_path, GT_path)."""

    def __init__(self, sizes):
        self.sizes = sizes

    def __len__(self):
        return len(self.sizes)

    def __getitem__(self, i):
        h, w = self.sizes[i]
        g = torch.Generator().manual_seed(i)
        return {"LQ": torch.rand(3, h, w, generator=g), "GT": torch.rand(3, 4 * h, 4 * w, generator=g),
                "LQ_path": "lq_%d.png" % i, "GT_path": "gt_%d.png" % i}


def test_batches_preserve_order_and_never_mix_shapes():
    sizes = [(8, 8)] * 5 + [(8, 12)] * 2 + [(8, 8)] * 1 + [(6, 8)] * 4
    ds = FakeSet(sizes)
    batches = list(batched_test_loader(ds, batch_size=3))
    assert [b["LQ"].shape[0] for b in batches] == [3, 2, 2, 1, 3, 1]
    seen = []
    for b in batches:
        assert b["LQ"].dim() == 4 and b["GT"].shape[2] == 4 * b["LQ"].shape[2]
        assert len(b["LQ_path"]) == b["LQ"].shape[0]
        seen += b["LQ_path"]
    assert seen == ["lq_%d.png" % i for i in range(len(sizes))]
    # contents: item i of the stream is dataset[i]
    k = 0
    for b in batches:
        for j in range(b["LQ"].shape[0]):
            assert torch.equal(b["LQ"][j], ds[k]["LQ"]) and torch.equal(b["GT"][j], ds[k]["GT"])
            k += 1
    # batch_size 1 reproduces the reference's loader
    assert [b["LQ"].shape[0] for b in batched_test_loader(ds, 1)] == [1] * len(sizes)
