"""ctypes binding of libhcflow_hip.so (the C ABI declared in include/hcflow.h).

There is no fallback: if the HIP library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HCFLOW_LIB", os.path.join(_HERE, "libhcflow_hip.so"))   # override: kernel experiments

HCF_OK = 0
ERR_NAMES = {-1: "HCF_ERR_ARG", -2: "HCF_ERR_HIP", -3: "HCF_ERR_STATE", -4: "HCF_ERR_KEY",
             -5: "HCF_ERR_SHAPE", -6: "HCF_ERR_UNSUPPORTED", -7: "HCF_ERR_NOMEM"}
FLAG_NO_CLAMP = 1
FLAG_NO_RANGE_CHECK = 2
FLAG_KEEP_COND = 4
FLAG_REUSE_COND = 8

# every symbol include/hcflow.h declares (tests/test_cabi_cpu.py checks the .so exports them all)
SYMBOLS = [
    "hcf_create", "hcf_destroy", "hcf_last_error", "hcf_param_count", "hcf_param_info",
    "hcf_set_param", "hcf_finalize", "hcf_inverse", "hcf_inverse_ex", "hcf_check_range", "hcf_check_range_samples", "hcf_aux_stream", "hcf_forward_sr", "hcf_forward_rescale",
    "hcf_workspace_bytes", "hcf_weight_bytes", "hcf_profile_convs", "hcf_conv_time_ms",
    "hcf_op_conv2d", "hcf_op_squeeze2d", "hcf_op_unsqueeze2d", "hcf_op_step_inverse",
    "hcf_op_step_forward_head", "hcf_op_step_forward_couple", "hcf_op_gauss_logp",
    "hcf_op_gauss_sample", "hcf_bench_conv", "hcf_set_precision", "hcf_get_precision", "hcf_fallback_count",
    "hcf_op_set_precision", "hcf_debug_set_ablation", "hcf_debug_last_clock_mhz", "hcf_debug_clock_probe",
    "hcf_actnorm_init_request", "hcf_get_param", "hcf_op_conv2d_backward",
    "hcf_train_forward_sr", "hcf_train_backward", "hcf_train_backward_phase", "hcf_bind_param_device", "hcf_refresh_from_device",
    "hcf_train_inverse", "hcf_train_backward_inverse", "hcf_metric_psnr_ssim", "hcf_metric_imresize_down",
    "hcf_train_select_tape", "hcf_train_forward_rescale", "hcf_train_backward_rescale",
    "hcf_debug_range_probe", "hcf_debug_range_probe_read",
    "hcf_aux_conv2d_workspace", "hcf_aux_conv2d", "hcf_aux_conv2d_backward", "hcf_adam_step",
]


class HcfError(RuntimeError):
    pass


class hcf_config(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("scale", C.c_int32), ("in_nc", C.c_int32), ("quant", C.c_float),
        ("L", C.c_int32), ("K", C.c_int32 * 4), ("after", C.c_int32 * 4), ("squeeze", C.c_int32),
        ("perm", C.c_int32), ("coupling", C.c_int32), ("nn_module", C.c_int32), ("hidden", C.c_int32),
        ("c_perm", C.c_int32), ("c_coupling", C.c_int32), ("c_nn_module", C.c_int32), ("c_hidden", C.c_int32),
        ("rrdb_nb", C.c_int32 * 2), ("rrdb_nf", C.c_int32), ("rrdb_gc", C.c_int32),
        ("lu_decomposed", C.c_int32),
    ]


_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HcfError(
            "libhcflow_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C hcflow_amd/csrc`). hcflow_amd has no CPU/PyTorch fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float
    fp = C.c_void_p          # device / host float pointers travel as raw addresses
    lib.hcf_create.argtypes = [C.POINTER(hcf_config), C.POINTER(vp)]
    lib.hcf_destroy.argtypes = [vp]
    lib.hcf_destroy.restype = None
    lib.hcf_last_error.argtypes = [vp]
    lib.hcf_last_error.restype = C.c_char_p
    lib.hcf_param_count.argtypes = [vp]
    lib.hcf_param_info.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(i32), C.POINTER(i64)]
    lib.hcf_set_param.argtypes = [vp, C.c_char_p, fp, C.POINTER(i64), i32]
    lib.hcf_finalize.argtypes = [vp, C.c_int]
    lib.hcf_inverse.argtypes = [vp, fp, C.POINTER(fp), i32, f32, u64, fp, i32, i32, i32, u32, vp]
    lib.hcf_inverse_ex.argtypes = [vp, fp, C.POINTER(fp), i32, f32, u64, i64, fp, i32, i32, i32, u32, vp]
    lib.hcf_check_range.argtypes = [vp, C.POINTER(i32)]
    lib.hcf_check_range_samples.argtypes = [vp, C.POINTER(i32), C.POINTER(u32)]
    lib.hcf_aux_stream.argtypes = [i32, i32, C.POINTER(vp)]
    lib.hcf_forward_sr.argtypes = [vp, fp, fp, fp, fp, fp, fp, fp, i32, i32, i32, vp]
    lib.hcf_forward_rescale.argtypes = [vp, fp, fp, fp, fp, i32, i32, i32, u32, vp]
    lib.hcf_workspace_bytes.argtypes = [vp]
    lib.hcf_workspace_bytes.restype = C.c_size_t
    lib.hcf_weight_bytes.argtypes = [vp]
    lib.hcf_weight_bytes.restype = C.c_size_t
    lib.hcf_set_precision.argtypes = [vp, i32]
    lib.hcf_get_precision.argtypes = [vp]
    lib.hcf_fallback_count.argtypes = [vp]
    lib.hcf_fallback_count.restype = C.c_int64
    lib.hcf_op_set_precision.argtypes = [i32]
    lib.hcf_debug_set_ablation.argtypes = [i32]
    lib.hcf_debug_last_clock_mhz.argtypes = []
    lib.hcf_debug_clock_probe.argtypes = [i32]
    lib.hcf_debug_last_clock_mhz.restype = C.c_double
    lib.hcf_bench_conv.argtypes = [i32, i32, i32, C.POINTER(i32), i32, i32, i32, i32, C.POINTER(C.c_double),
                                   C.POINTER(C.c_double), vp]
    lib.hcf_profile_convs.argtypes = [vp, C.c_int]
    lib.hcf_conv_time_ms.argtypes = [vp, i32, i32, i32, i32, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]
    lib.hcf_metric_psnr_ssim.argtypes = [fp, fp, i32, i32, i32, i32, i32, fp, vp]
    lib.hcf_metric_imresize_down.argtypes = [fp, i32, i32, i32, i32, fp, vp]
    lib.hcf_train_inverse.argtypes = [vp, fp, C.POINTER(fp), i32, f32, u64, fp, i32, i32, i32, C.c_uint32, vp]
    lib.hcf_train_backward_inverse.argtypes = [vp, fp, fp, i64, fp, vp]
    lib.hcf_train_select_tape.argtypes = [vp, i32]
    lib.hcf_train_forward_rescale.argtypes = [vp, fp, fp, fp, fp, i32, i32, i32, C.c_uint32, vp]
    lib.hcf_train_backward_rescale.argtypes = [vp, fp, fp, fp, fp, i64, vp]
    lib.hcf_adam_step.argtypes = [vp, vp, vp, vp, i32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, i64, vp]
    lib.hcf_aux_conv2d_workspace.argtypes = [i32, i32, i32, i32, i32, i32]
    lib.hcf_aux_conv2d_workspace.restype = C.c_size_t
    lib.hcf_aux_conv2d.argtypes = [fp, i32, i32, i32, i32, i32, fp, fp, i32, i32, i32, fp, i32, vp, C.c_size_t, i32, vp]
    lib.hcf_aux_conv2d_backward.argtypes = [fp, i32, i32, i32, i32, i32, fp, i32, i32, fp, i32, fp, i32, fp, vp, C.c_size_t, i32, vp]
    lib.hcf_debug_range_probe.argtypes = [vp, i32]
    lib.hcf_debug_range_probe_read.argtypes = [vp, i32, C.c_char_p, i32, C.POINTER(f32), C.POINTER(i32)]
    lib.hcf_bind_param_device.argtypes = [vp, C.c_char_p, fp]
    lib.hcf_refresh_from_device.argtypes = [vp, vp]
    lib.hcf_train_forward_sr.argtypes = [vp, fp, fp, fp, fp, fp, fp, i32, i32, i32, vp]
    lib.hcf_train_backward.argtypes = [vp, f32, fp, i64, vp]
    lib.hcf_train_backward_phase.argtypes = [vp, i32, f32, fp, i64, vp]
    lib.hcf_actnorm_init_request.argtypes = [vp, C.POINTER(C.c_char_p), i32]
    lib.hcf_get_param.argtypes = [vp, C.c_char_p, fp, i64]
    lib.hcf_op_conv2d.argtypes = [C.POINTER(fp), C.POINTER(i32), C.POINTER(i32), i32, i32, i32, i32, fp, fp, fp,
                                  i32, i32, i32, fp, f32, fp, f32, fp, vp]
    lib.hcf_op_conv2d_backward.argtypes = [C.POINTER(fp), C.POINTER(i32), C.POINTER(i32), i32, i32, i32, i32, fp, i32, i32,
                                           fp, C.POINTER(fp), fp, fp, vp]
    lib.hcf_op_squeeze2d.argtypes = [fp, fp, i32, i32, i32, i32, i32, vp]
    lib.hcf_op_unsqueeze2d.argtypes = [fp, fp, i32, i32, i32, i32, i32, vp]
    lib.hcf_op_step_inverse.argtypes = [fp, fp, fp, i32, i32, i32, i32, i32, i32, i32, fp, fp, fp, vp]
    lib.hcf_op_step_forward_head.argtypes = [fp, fp, i32, i32, i32, i32, fp, fp, fp, vp]
    lib.hcf_op_step_forward_couple.argtypes = [fp, fp, fp, fp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.hcf_op_gauss_logp.argtypes = [fp, fp, fp, i32, i32, i32, i32, vp]
    lib.hcf_op_gauss_sample.argtypes = [fp, fp, f32, u64, fp, i32, i32, i32, i32, i32, vp]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("hcf_destroy", "hcf_last_error", "hcf_workspace_bytes", "hcf_weight_bytes",
                        "hcf_fallback_count", "hcf_debug_last_clock_mhz", "hcf_aux_conv2d_workspace"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def check(rc: int, engine=None, what: str = ""):
    if rc == HCF_OK:
        return
    msg = ""
    if engine is not None:
        m = load().hcf_last_error(engine)
        msg = m.decode() if m else ""
    raise HcfError("%s failed: %s (%d) %s" % (what or "hcflow call", ERR_NAMES.get(rc, "?"), rc, msg))


def make_config(cfg) -> hcf_config:
    """hcflow_amd.config.NetConfig -> C struct."""
    c = hcf_config()
    c.kind = 0 if cfg.sr else 1
    c.scale = cfg.scale
    c.in_nc = cfg.in_nc
    c.quant = float(cfg.quant)
    c.L = cfg.L
    for i in range(4):
        c.K[i] = cfg.K[i] if i < len(cfg.K) else 0
        c.after[i] = cfg.after[i] if i < len(cfg.after) else 0
    c.squeeze = 1 if cfg.squeeze == "haar" else 0
    perm = {"invconv": 0, "none": 1}
    cpl = {"Affine": 0, "Affine3shift": 1}
    nn = {"FCN": 0, "DenseBlock": 1}
    c.perm, c.coupling, c.nn_module, c.hidden = perm[cfg.perm], cpl[cfg.coupling], nn[cfg.nn_module], cfg.hidden
    c.c_perm, c.c_coupling, c.c_nn_module, c.c_hidden = (perm[cfg.c_perm], cpl[cfg.c_coupling],
                                                         nn[cfg.c_nn_module], cfg.c_hidden)
    c.rrdb_nb[0], c.rrdb_nb[1] = cfg.rrdb_nb
    c.rrdb_nf, c.rrdb_gc = cfg.rrdb_nf, cfg.rrdb_gc
    c.lu_decomposed = 1 if getattr(cfg, "lu", False) else 0
    return c


class Engine:
    """Owning handle around hcf_engine*."""

    def __init__(self, cfg):
        self.lib = load()
        self.cfg = cfg
        self._h = C.c_void_p()
        c = make_config(cfg)
        check(self.lib.hcf_create(C.byref(c), C.byref(self._h)), None, "hcf_create")

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h:
                self.lib.hcf_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def param_spec(self) -> List[Tuple[str, Tuple[int, ...]]]:
        n = self.lib.hcf_param_count(self._h)
        out = []
        key = C.c_char_p()
        nd = C.c_int32()
        shape = (C.c_int64 * 4)()
        for i in range(n):
            check(self.lib.hcf_param_info(self._h, i, C.byref(key), C.byref(nd), shape), self._h, "hcf_param_info")
            out.append((key.value.decode(), tuple(int(shape[j]) for j in range(nd.value))))
        return out

    def set_param(self, key: str, host_tensor):
        """host_tensor: contiguous fp32 CPU torch tensor."""
        t = host_tensor
        assert t.device.type == "cpu" and t.is_contiguous() and str(t.dtype) == "torch.float32", key
        shape = (C.c_int64 * 4)(*([int(s) for s in t.shape] + [1] * (4 - t.dim())))
        check(self.lib.hcf_set_param(self._h, key.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()), self._h,
              "hcf_set_param(%s)" % key)

    def actnorm_init_request(self, prefixes):
        """Arm the next forward pass to fit the listed ActNorms from its data (include/hcflow.h)."""
        arr = (C.c_char_p * len(prefixes))(*[p.encode() for p in prefixes])
        check(self.lib.hcf_actnorm_init_request(self._h, arr, len(prefixes)), self._h, "hcf_actnorm_init_request")

    def get_param(self, key: str, numel: int):
        """The engine's host copy of a parameter as a flat fp32 CPU tensor."""
        import torch
        out = torch.empty(int(numel), dtype=torch.float32)
        check(self.lib.hcf_get_param(self._h, key.encode(), C.c_void_p(out.data_ptr()), int(numel)), self._h,
              "hcf_get_param(%s)" % key)
        return out

    def bind_param_device(self, key: str, data_ptr: int):
        check(self.lib.hcf_bind_param_device(self._h, key.encode(), C.c_void_p(data_ptr)), self._h, "hcf_bind_param_device")

    def refresh_from_device(self, stream):
        check(self.lib.hcf_refresh_from_device(self._h, stream), self._h, "hcf_refresh_from_device")

    def finalize(self, device: int):
        check(self.lib.hcf_finalize(self._h, int(device)), self._h, "hcf_finalize")

    PRECISIONS = {"exact": 0, "f16x3": 1}

    def set_precision(self, mode: str):
        check(self.lib.hcf_set_precision(self._h, self.PRECISIONS[mode]), self._h, "hcf_set_precision")

    def fallback_count(self) -> int:
        return int(self.lib.hcf_fallback_count(self._h))

    def check_range(self) -> bool:
        """True when an f16x3 pass since the last check saw an input beyond the f16 range (include/hcflow.h: hcf_check_range).
        Waits for the passes enqueued so far."""
        o = C.c_int32(0)
        check(self.lib.hcf_check_range(self._h, C.byref(o)), self._h, "hcf_check_range")
        return bool(o.value)

    def check_range_samples(self):
        """(overflowed, slots): as check_range, plus the bit set of flagged sample slots -- bit (b mod 30) for sample b of its call
        (include/hcflow.h: hcf_check_range_samples)."""
        o, m = C.c_int32(0), C.c_uint32(0)
        check(self.lib.hcf_check_range_samples(self._h, C.byref(o), C.byref(m)), self._h, "hcf_check_range_samples")
        return bool(o.value), int(m.value)

    def range_probe(self, enable: bool):
        check(self.lib.hcf_debug_range_probe(self._h, int(enable)), self._h, "hcf_debug_range_probe")

    def range_probe_records(self):
        """[(weight key, max|x|, max|V|, cin, cout, H, W, ran_f16x3, has_wino)] in launch order since range_probe(True)."""
        out, i = [], 0
        key = C.create_string_buffer(256)
        mx = (C.c_float * 2)()
        info = (C.c_int32 * 6)()
        while True:
            rc = self.lib.hcf_debug_range_probe_read(self._h, i, key, 256, mx, info)
            if rc == -4:
                break
            check(rc, self._h, "hcf_debug_range_probe_read")
            out.append((key.value.decode(), float(mx[0]), float(mx[1])) + tuple(int(v) for v in info))
            i += 1
        return out

    def profile_convs(self, enable: bool):
        check(self.lib.hcf_profile_convs(self._h, int(enable)), self._h, "hcf_profile_convs")

    def conv_time(self, taps: int = 0, nt: int = 0, reset: bool = False, kind: int = -1):
        """(total_ms, launches, algorithmic_flops, algorithmic_bytes) of the recorded conv launches of one variant
        (kind: 0 plain, 1 fused 1x1 second layer, 2 fused flow-step tail, 3 upsampled source, 4 Winograd form, 5 persistent small-K FCN kernel,
        6 Winograd conv1 + 1x1 conv2 of a conditional FCN, 7 completion of a fat dense-block launch, -1 any)."""
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        check(self.lib.hcf_conv_time_ms(self._h, taps, nt, kind, int(reset), C.byref(ms), C.byref(n), C.byref(fl),
                                        C.byref(by)), self._h, "hcf_conv_time_ms")
        return ms.value, n.value, fl.value, by.value

    def workspace_bytes(self) -> int:
        return int(self.lib.hcf_workspace_bytes(self._h))

    def weight_bytes(self) -> int:
        return int(self.lib.hcf_weight_bytes(self._h))
