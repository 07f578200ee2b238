# Drop-in replacement for codes/models/modules/HCFlowNet_SR_arch.py of JingyunLiang/HCFlow.
# networks.find_model_using_name (codes/models/networks.py:9-25) imports this module and picks the
# attribute whose lower-cased name equals which_model_G ("HCFlowNet_SR").
from hcflow_amd.arch import HCFlowNet_SR  # noqa: F401
