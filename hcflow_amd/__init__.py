"""hcflow_amd -- MI355X-native HCFlow forward / inverse engine.

Host side (this package) mirrors the reference's arch-module interface; the compute lives in
hand-written HIP kernels behind the C ABI of include/hcflow.h (hcflow_amd/libhcflow_hip.so).
"""
from .config import NetConfig, preset, param_spec, eps_shapes  # noqa: F401
from .params import make_params  # noqa: F401
from .arch import HCFlowNet_SR, HCFlowNet_Rescaling  # noqa: F401

__all__ = ["NetConfig", "preset", "param_spec", "eps_shapes", "make_params", "HCFlowNet_SR", "HCFlowNet_Rescaling"]
