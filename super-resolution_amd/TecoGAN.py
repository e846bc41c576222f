"""Temporally Coherent GAN generator on MI355X: drop-in for the generator side of reference TecoGAN.py.

``generator`` = the FRVSR graph plus ``tanh(SRNet(...) + bilinear_x4(I_LR_t))`` (TecoGAN.py:65-80;
the reference's ``bicubic_upscale`` is in fact bilinear, layers.py:89) and ``hr_warping`` returning the
warped previous estimate for the same three inputs (TecoGAN.py:83-85).  ``train()`` runs the generator
pre-training phase (TecoGAN.py:164-275: the generator compiled with two MSE losses, learning rate halved
every 50k iterations down to 2.5e-5) on the native training step.  The discriminator, the VGG perceptual
losses and the adversarial phase (TecoGAN.py:82-131, 277-527) are the "next" row of the scope table
(SURVEY.md section 8 f rank 3) and are not built here.
"""
from FRVSR import FrameRecurrentVideoSR
from layers import Adam, GeneratorModel, HRWarpingModel, Upscaling


class TecoGAN(FrameRecurrentVideoSR):
    model_kind = 'tecogan'

    def build(self):
        super().build()
        self.bicubic_upscale = Upscaling(self.lr_shape, 1, device=self.core.device).build()
        self.generator = GeneratorModel(self.core, 'Generator')
        self.hr_warping = HRWarpingModel(self.core)
        self.opt = Adam(lr=5e-5)                                    # TecoGAN.py:110-113
        self.generator.compile(loss=['mean_squared_error', 'mean_squared_error'], loss_weights=[1., 1.], optimizer=self.opt)
        del self.frvsr
        print('[INFO] Temporally Coherent GAN ready.')

    def _generator(self):
        return self.generator

    @staticmethod
    def gen_learning_rate(i):
        """1e-4 halved every 50k iterations, floor 2.5e-5 (TecoGAN.py:173-175, 225-228)."""
        return max(1e-4 / 2 ** (i // 5e4), 2.5e-5)

    def train(self):
        """Generator pre-training: iterations i < train.iterations / 2 (TecoGAN.py:178), skipped like in the
        reference when a pretrained discriminator is configured; then the adversarial phase, which is not built."""
        cfg = self.configs['train']
        i = self._start_iteration('[INFO] Generator ready.')
        if not cfg.get('pretrained_disc'):
            print('[INFO] Starting generator training process ...')
            self._train_generator(i, cfg['iterations'] / 2, self.gen_learning_rate,
                                  'generator_model_weights_%d.h5', 'TecoGAN.yaml', 'Generator')
        raise NotImplementedError(
            'TecoGAN adversarial phase (reference TecoGAN.py:277-527) needs the discriminator and VGG19 ImageNet '
            'weights (a download): next row of SURVEY.md section 8 f; the generator pre-training phase above is done')
