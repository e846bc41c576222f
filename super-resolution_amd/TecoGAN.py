"""Temporally Coherent GAN generator on MI355X: drop-in for the generator side of reference TecoGAN.py.

``generator`` = the FRVSR graph plus ``tanh(SRNet(...) + bilinear_x4(I_LR_t))`` (TecoGAN.py:65-80;
the reference's ``bicubic_upscale`` is in fact bilinear, layers.py:89) and ``hr_warping`` returning the
warped previous estimate for the same three inputs (TecoGAN.py:83-85).  The discriminator, the VGG
perceptual losses and the adversarial training loop (TecoGAN.py:82-131, 136-527) are the "next" rows of
the scope table (SURVEY.md section 8 f) and are not built here.
"""
from FRVSR import FrameRecurrentVideoSR
from layers import GeneratorModel, HRWarpingModel, Upscaling


class TecoGAN(FrameRecurrentVideoSR):
    model_kind = 'tecogan'

    def build(self):
        super().build()
        self.bicubic_upscale = Upscaling(self.lr_shape, 1).build()
        self.generator = GeneratorModel(self.core, 'Generator')
        self.hr_warping = HRWarpingModel(self.core)
        del self.frvsr
        print('[INFO] Temporally Coherent GAN ready.')

    def _generator(self):
        return self.generator

    def train(self):
        raise NotImplementedError(
            'TecoGAN adversarial training (reference TecoGAN.py:136-527) needs the discriminator, VGG19 '
            'ImageNet weights (a download) and backward kernels: next rows of SURVEY.md section 8 f')
