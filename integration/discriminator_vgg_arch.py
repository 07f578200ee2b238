# Drop-in replacement for codes/models/modules/discriminator_vgg_arch.py of JingyunLiang/HCFlow (HCFlow+ / ++ recipes):
# networks.define_D / define_F (codes/models/networks.py:44-72) look these two classes up by attribute. Discriminator_VGG_128
# and PatchGANDiscriminator are not used by any shipped yml (train_SR_*_HCFlow++.yml: which_model_D: discriminator_vgg_160).
from hcflow_amd.gan import Discriminator_VGG_160, VGGFeatureExtractor  # noqa: F401
