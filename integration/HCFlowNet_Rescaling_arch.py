# Drop-in replacement for codes/models/modules/HCFlowNet_Rescaling_arch.py of JingyunLiang/HCFlow.
from hcflow_amd.arch import HCFlowNet_Rescaling  # noqa: F401
