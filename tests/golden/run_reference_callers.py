#!/usr/bin/env python
"""Run the REFERENCE'S OWN CALLERS on top of the drop-in classes (build container only; CPU; copies nothing).

    python tests/golden/run_reference_callers.py        # writes tests/golden/callers_*.npz

What is observed here, not argued (SURVEY.md 8b / INTEGRATION.md):

* the reference's option parser (`codes/options/options.py: parse, dict_to_nonedict`) on the reference's real yml files;
* `models.create_model` -> `HCFlowSRModel.__init__` / `HCFLowRescalingModel.__init__` (HCFlow_SR_model.py:19-158,
  HCFlow_Rescaling_model.py) -> `networks.define_G` (networks.py:36-41) finds OUR class through the same importlib lookup,
  because `integration/*_arch.py` stands where the reference's arch files would be (sys.modules entry = "the file was replaced");
* `.to(device)`, `DataParallel(netG)`, `print_network`, `load()` -> `load_network(strict=True)` of a checkpoint WRITTEN BY THE
  REFERENCE'S OWN MODULE through `BaseModel.save_network` (base_model.py:79-120), `set_actnorm_init`; and the reverse direction:
  `save_network(our netG)` -> `load_network` into the reference's module, strict;
* the optimizer parameter groups (`HCFlow_SR_model.py:104-125`), schedulers, `feed_data`, `test()` and
  `optimize_parameters(step)` with every keyword argument they pass to `netG(...)`.

There is no GPU in the build container and the classes have no CPU path (they raise `HcfError`), so the arithmetic of each
recorded `netG(...)` call is DELEGATED to the reference's own module sharing the same Parameter objects -- that lets the callers
run to the end (losses, backward, gradient clip, Adam step, metrics loop) and yields the reference's outputs for exactly the
calls they make. The recorded kwargs, random draws and outputs are committed as fixtures and replayed through the real engine on
the GPU box (tests/test_gpu_callers.py); tests/test_callers_cpu.py checks the recorded call surface against our signature.

Stubs: cv2 / natsort / lpips / torchvision / tensorboard are absent from this image and are imported by the reference at module
level only; they are replaced by empty modules (natsort.natsorted = sorted). Nothing of the reference is copied or modified.
"""
import collections
import importlib
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/codes"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import Capture, np_, pack_out, img_tensor  # noqa: E402
from hcflow_amd.params import make_params  # noqa: E402
from hcflow_amd.config import NetConfig  # noqa: E402
from hcflow_amd import arch as our_arch  # noqa: E402
from hcflow_amd._lib import HcfError  # noqa: E402


def stub_absent_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    mod("cv2")
    mod("natsort", natsorted=lambda seq, **k: sorted(seq, reverse=k.get("reverse", False)))
    mod("lpips")
    tv = mod("torchvision")
    tv.utils = mod("torchvision.utils", make_grid=None)
    tv.models = mod("torchvision.models")
    mod("tensorboard")


def load_file_as(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def install():
    """Reference on sys.path, the two arch files 'replaced' by integration/*.py, the originals importable under other names."""
    stub_absent_modules()
    sys.path.insert(0, REF)
    import models.modules  # noqa: F401  (the reference's package)
    ref_sr = load_file_as("ref_HCFlowNet_SR_arch", os.path.join(REF, "models/modules/HCFlowNet_SR_arch.py")).HCFlowNet_SR
    ref_rs = load_file_as("ref_HCFlowNet_Rescaling_arch",
                          os.path.join(REF, "models/modules/HCFlowNet_Rescaling_arch.py")).HCFlowNet_Rescaling
    load_file_as("models.modules.HCFlowNet_SR_arch", os.path.join(ROOT, "integration/HCFlowNet_SR_arch.py"))
    load_file_as("models.modules.HCFlowNet_Rescaling_arch", os.path.join(ROOT, "integration/HCFlowNet_Rescaling_arch.py"))
    return ref_sr, ref_rs


class Recorder:
    """Wraps forward of OUR class: records the keyword arguments of every call, checks that the class itself refuses to compute
    on the CPU (HcfError: no fallback), then delegates the arithmetic to the reference module that shares its Parameters."""

    def __init__(self):
        self.calls = []
        self.delegate = {}
        self.cap = None

    def install(self, cls):
        rec = self
        orig = cls.forward

        def forward(self_, *args, **kw):
            assert not args, "the reference's callers pass keyword arguments only"
            ent = {"cls": type(self_).__name__, "training_mode": bool(self_.training), "grad": torch.is_grad_enabled(), "kw": {}}
            for k, v in kw.items():
                ent["kw"][k] = ("tensor", tuple(v.shape), str(v.dtype), bool(v.requires_grad)) if torch.is_tensor(v) else ("value", v)
            rec.calls.append(ent)
            try:                                           # the product path has no CPU fallback
                with torch.no_grad():
                    orig(self_, *args, **kw)
                raise AssertionError("hcflow_amd computed something without a GPU")
            except (HcfError, NotImplementedError) as e:
                ent["cpu_refusal"] = type(e).__name__
            ref = rec.delegate[id(self_)]
            mods = dict(ref.named_modules())
            for name, m in self_.named_modules():          # ActNorm flags live on OUR modules (set_actnorm_init)
                if "ActNorm" in type(m).__name__:
                    mods[name].inited = m.inited
            ref.train(self_.training)
            out = ref(**kw)
            for name, m in self_.named_modules():
                if "ActNorm" in type(m).__name__:
                    m.inited = mods[name].inited
            ent["out"] = out
            return out
        cls.forward = forward


class DrawLog(Capture):
    """Capture that also logs, in call order, what was drawn: ('rand', shape) / ('normal', shape, std). The draws come from
    torch's global CPU generator after torch.manual_seed(seed) (test_HCFlow.py:34 util.set_random_seed(0)), so the GPU box
    regenerates them with the same calls (tests/util.py::regen_draws) instead of shipping megabytes of noise; digests check it."""

    def __enter__(self):
        self.order = []
        super().__enter__()
        n_, r_ = torch.normal, torch.rand

        def normal(*a, **k):
            o = n_(*a, **k)
            std = k["std"] if "std" in k else a[1]
            self.order.append(("normal", tuple(o.shape), float(std.flatten()[0]) if torch.is_tensor(std) else float(std)))
            return o

        def rand(*a, **k):
            o = r_(*a, **k)
            self.order.append(("rand", tuple(o.shape), 1.0))
            return o
        torch.normal, torch.rand = normal, rand
        return self


def store_draws(out, tag, cap):
    draws = {"normal": list(cap.normal), "rand": list(cap.rand)}
    rows, dig = [], []
    for kind, shape, std in cap.order:
        e = draws[kind].pop(0)
        assert tuple(e.shape) == shape
        rows.append("%s|%s|%r" % (kind, ",".join(str(v) for v in shape), std))
        dig.append([float(e.double().sum()), float((e.double() ** 2).sum())])
    out[tag + "_draws"] = np.array(rows)
    out[tag + "_draw_digest"] = np.array(dig, dtype=np.float64).reshape(-1, 2)


def share_parameters(ours, ref):
    """The reference module computes with OUR Parameter objects (same names: the state-dict tables are identical)."""
    ref_mods = dict(ref.named_modules())
    n = 0
    for key, p in ours.named_parameters():
        path, name = key.rsplit(".", 1)
        m = ref_mods[path]
        assert name in m._parameters and tuple(m._parameters[name].shape) == tuple(p.shape), key
        m._parameters[name] = p
        n += 1
    assert n == len(list(ref.parameters()))


def shrink(opt, K, after, nb):
    """Reduced depth for the CPU run of the TRAINING callers (same widths): bounded time, everything else as shipped."""
    fd = opt["network_G"]["flowDownsampler"]
    fd["K"] = K
    fd["splitOff"]["after_flowstep"] = after
    fd["splitOff"]["RRDB_nb"] = nb


def seeded_state(ref_cls, opt, seed):
    """A reference module holding the seeded recipe (a fresh reference net is a near-identity, hcflow_amd/params.py)."""
    net = ref_cls(opt=opt, step=0)
    cfg = NetConfig.from_opt(opt)
    net.load_state_dict(make_params(cfg, seed), strict=True)
    return net, cfg


def kw_table(calls):
    rows = []
    for c in calls:
        items = []
        for k, v in c["kw"].items():
            items.append("%s=%s" % (k, "T%s" % (list(v[1]),) if v[0] == "tensor" else repr(v[1])))
        rows.append("%s|train=%d|grad=%d|%s|%s" % (c["cls"], c["training_mode"], c["grad"], c.get("cpu_refusal", "-"), ", ".join(items)))
    return rows


def run_test_driver(tag, yml, ref_cls, rec, lr, hr, seed, out):
    """test_HCFlow.py:18-21,47-48,85-90: parse -> create_model -> feed_data -> test() -> get_current_visuals."""
    import options.options as option
    from models import create_model
    opt = option.parse(os.path.join(REF, "options/test", yml), is_train=False)
    opt = option.dict_to_nonedict(opt)
    opt["gpu_ids"] = None                                  # BaseModel: device = cpu (no GPU in the build container)
    opt["dist"] = False
    tmp = tempfile.mkdtemp(prefix="hcf_callers_")
    opt["path"]["models"] = tmp
    ref_net, cfg = seeded_state(ref_cls, opt, seed)
    # (1) the REFERENCE writes the checkpoint with its own saver, from its own module under DataParallel
    from models.base_model import BaseModel
    saver = BaseModel(opt)
    saver.save_network(torch.nn.DataParallel(ref_net), "G", "ref")
    ckpt = os.path.join(tmp, "ref_G.pth")
    assert os.path.exists(ckpt)
    opt["path"]["pretrain_model_G"] = ckpt
    n0 = len(rec.calls)
    model = create_model(opt)                              # __init__ -> define_G -> OUR class; load() -> load_network strict
    net = model.netG.module
    assert type(net).__module__ == "hcflow_amd.arch", type(net)
    assert all(m.inited for m in net.modules() if "ActNorm" in type(m).__name__)        # load(): set_actnorm_init(True)
    for (ka, a), (kb, b) in zip(net.state_dict().items(), ref_net.state_dict().items()):
        assert ka == kb and torch.equal(a, b), ka
    # (2) and back: OUR module through the reference's saver, into the reference's module, strict
    model.save_network(model.netG, "G", "ours")
    ref_back = ref_cls(opt=opt, step=0)
    model.load_network(os.path.join(tmp, "ours_G.pth"), torch.nn.DataParallel(ref_back), strict=True)
    assert all(torch.equal(a, b) for a, b in zip(ref_back.state_dict().values(), ref_net.state_dict().values()))
    share_parameters(net, ref_net)
    rec.delegate[id(net)] = ref_net
    model.feed_data({"LQ": lr, "GT": hr}, need_GT=True)
    torch.manual_seed(0)                                   # util.set_random_seed(0), test_HCFlow.py:34
    with DrawLog() as cap:
        ret = model.test()
    vis = model.get_current_visuals(need_GT=True)
    calls = rec.calls[n0:]
    out[tag + "_calls"] = np.array(kw_table(calls))
    out[tag + "_seed"] = seed
    out[tag + "_preset_opt"] = np.array([repr(cfg)])
    out[tag + "_heats"] = np.array([float(h) for h in opt["val"]["heats"]])
    out[tag + "_n_sample"] = int(opt["val"]["n_sample"])
    out[tag + "_test_return"] = np.float64(ret)
    store_draws(out, tag, cap)
    out[tag + "_lq_fromH"] = np_(vis["LQ_fromH"])
    for (kind, heat, i), v in [(k, v) for k, v in vis.items() if isinstance(k, tuple)]:
        pack_out(out, "%s_SR_%g_%d" % (tag, heat, i), v)
    print("  %s: %d netG calls, test() -> %.6f, %d normal / %d rand draws" % (tag, len(calls), ret, len(cap.normal), len(cap.rand)))
    for r in kw_table(calls):
        print("     ", r)
    return cfg


def run_train_driver(tag, yml, ref_cls, rec, seed, out, steps=2):
    """train_HCFlow.py main loop body: create_model (is_train) -> feed_data -> optimize_parameters(step) ->
    update_learning_rate -> get_current_log, at reduced depth / batch 2."""
    import options.options as option
    from models import create_model
    opt = option.parse(os.path.join(REF, "options/train", yml), is_train=True)
    opt = option.dict_to_nonedict(opt)
    opt["gpu_ids"] = None
    opt["dist"] = False
    sr = opt["model"] == "HCFlow_SR"
    shrink(opt, 4 if sr else 5, [2, 2], [1, 1])
    opt["path"]["pretrain_model_G"] = None
    opt["path"]["resume_state"] = None
    ref_net, cfg = seeded_state(ref_cls, opt, seed)
    with torch.no_grad():                                  # a fresh training run: ActNorms still to be fitted (zero bias / logs)
        for k, v in ref_net.state_dict().items():
            if ".actnorm." in k:
                v.zero_()
    n0 = len(rec.calls)
    model = create_model(opt)
    net = model.netG.module
    assert type(net).__module__ == "hcflow_amd.arch", type(net)
    net.load_state_dict(ref_net.state_dict(), strict=True)
    share_parameters(net, ref_net)
    rec.delegate[id(net)] = ref_net
    # optimizer parameter groups (HCFlow_SR_model.py:104-125): every requires_grad parameter of netG, in named_parameters order
    group = model.optimizer_G.param_groups[0]
    names = {id(p): k for k, p in net.named_parameters()}
    out[tag + "_optim_keys"] = np.array([names[id(p)] for p in group["params"]])
    out[tag + "_optim_hyper"] = np.array([group["lr"], group["betas"][0], group["betas"][1], group["weight_decay"], group["eps"]])
    if sr:
        out[tag + "_recipe"] = np.array([float(model.l_nll_w)])
    else:
        out[tag + "_recipe"] = np.array([float(model.l_pix_w_lr), float(model.l_w_z), float(model.l_pix_w_hr), float(model.eps_std_reverse)])
        out[tag + "_criteria"] = np.array([type(model.cri_pix_lr).__name__, type(model.cri_pix_hr).__name__])
    out[tag + "_act_norm_start_step"] = int(opt["network_G"]["act_norm_start_step"])
    out[tag + "_preset_opt"] = np.array([repr(NetConfig.from_opt(opt))])
    out[tag + "_max_grad"] = np.array([float(model.max_grad_clip or 0), float(model.max_grad_norm or 0)])
    g = torch.Generator().manual_seed(seed + 7)
    s = cfg.scale
    hr = torch.rand(2, 3, 12 * s, 10 * s, generator=g) * 0.8 + 0.1
    lr = torch.nn.functional.avg_pool2d(hr, s)
    out[tag + "_hr"], out[tag + "_lr"] = np_(hr), np_(lr)
    out[tag + "_seed"] = seed
    model.feed_data({"LQ": lr, "GT": hr})
    logs = []
    torch.manual_seed(0)                                   # util.set_random_seed(seed), train_HCFlow.py
    with DrawLog() as cap:
        for step in range(steps):
            model.optimize_parameters(step)
            model.update_learning_rate(step, warmup_iter=opt["train"]["warmup_iter"] or -1)
            logs.append(dict(model.get_current_log()))
            if step == 0:                                  # ActNorms fitted by the reference inside step 0
                an = [(k, m) for k, m in net.named_modules() if "ActNorm" in type(m).__name__]
                assert all(m.inited for _, m in an)
    calls = rec.calls[n0:]
    out[tag + "_calls"] = np.array(kw_table(calls))
    out[tag + "_log_keys"] = np.array(sorted(logs[0].keys()))
    out[tag + "_logs"] = np.array([[float(lg[k]) for k in sorted(lg.keys())] for lg in logs])
    store_draws(out, tag, cap)
    # parameters after `steps` optimiser steps: digest per tensor (l2, sum) -- the replay on the GPU runs the same recipe
    out[tag + "_param_digest"] = np.array([[float(v.double().norm()), float(v.double().sum())] for v in net.state_dict().values()])
    print("  %s: %d netG calls over %d optimize_parameters steps, logs %s" % (tag, len(calls), steps, logs))
    for r in kw_table(calls):
        print("     ", r)


def main():
    torch.set_num_threads(8)
    ref_sr, ref_rs = install()
    rec = Recorder()
    rec.install(our_arch.HCFlowNet_SR)
    rec.install(our_arch.HCFlowNet_Rescaling)
    im = np.load(os.path.join(HERE, "real_images.npz"))
    out = {}
    run_test_driver("test_sr4", "test_SR_DF2K_4X_HCFlow.yml", ref_sr, rec, img_tensor(im["butterfly_lr"]),
                    img_tensor(im["butterfly_hr"]), 91, out)
    run_test_driver("test_sr8", "test_SR_CelebA_8X_HCFlow.yml", ref_sr, rec, img_tensor(im["face_lr"][:1]),
                    img_tensor(im["face_hr"][:1]), 92, out)
    run_test_driver("test_rescale", "test_Rescaling_DF2K_4X_HCFlow.yml", ref_rs, rec, img_tensor(im["butterfly_lr"]),
                    img_tensor(im["butterfly_hr"]), 93, out)
    np.savez_compressed(os.path.join(HERE, "callers_test.npz"), **out)
    print("wrote callers_test.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "callers_test.npz")) / 1024))
    out = {}
    run_train_driver("train_sr4", "train_SR_DF2K_4X_HCFlow.yml", ref_sr, rec, 94, out)
    run_train_driver("train_rescale", "train_Rescaling_DF2K_4X_HCFlow.yml", ref_rs, rec, 95, out)
    np.savez_compressed(os.path.join(HERE, "callers_train.npz"), **out)
    print("wrote callers_train.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "callers_train.npz")) / 1024))


if __name__ == "__main__":
    main()
