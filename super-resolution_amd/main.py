"""Entry point with the reference's CLI and YAML surface (reference main.py:21-85).

    python main.py --yaml_file options/FRVSR_example.yaml

Same keys as the reference (root_dir, stage.{train,eval,run}, gpu, data.{low_res,high_res,channels,
upscale,videos,rows,cols}, cnn.{model,pretrained_model}, eval.output_dir); extra optional keys:
``dtype`` (fp32 | bf16x3 | f16mx6 | bf16) and ``data.low_res_w`` / ``data.high_res_w`` for non-square tiles.
For several GPUs launch one process per GPU:
    python -m torch.distributed.run --nproc-per-node N main.py --yaml_file X.yaml
each rank evaluates a contiguous block of the clips (no collective: clips are independent).
"""
import argparse
import datetime
import os
import sys

import yaml

from FRVSR import FrameRecurrentVideoSR
from TecoGAN import TecoGAN


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description='Video Super-Resolution')
    ap.add_argument('--yaml_file', help='YAML file.')
    return ap.parse_args(argv)


def shapes_from_config(configs):
    d = configs['data']
    lr_h, hr_h = d['low_res'], d['high_res']
    lr_w, hr_w = d.get('low_res_w', lr_h), d.get('high_res_w', hr_h)
    lr_shape, hr_shape = (lr_h, lr_w, d['channels']), (hr_h, hr_w, d['channels'])
    if lr_shape[0] * d['upscale'] != hr_shape[0] or lr_shape[1] * d['upscale'] != hr_shape[1]:
        raise ValueError('[ERROR] Incompatible resolutions with upscale x%d.' % d['upscale'])
    return lr_shape, hr_shape


def make_model(configs, output_dir):
    lr_shape, hr_shape = shapes_from_config(configs)
    kind = configs['cnn']['model']
    if kind == 'frvsr':
        return FrameRecurrentVideoSR(lr_shape, hr_shape, output_dir, configs)
    if kind == 'tecogan':
        return TecoGAN(lr_shape, hr_shape, output_dir, configs)
    raise ValueError('[ERROR] Not valid model selected.')


def main(argv=None):
    args = parse_args(argv)
    with open(args.yaml_file) as fh:
        configs = yaml.safe_load(fh)
    now = datetime.datetime.now()
    output_dir = os.path.join(configs['root_dir'], 'MODEL', '%d-%d-%d %d:%d:%d' % (
        now.year, now.month, now.day, now.hour, now.minute, now.second))
    stage = configs['stage']
    try:
        model = make_model(configs, output_dir)
    except ValueError as e:
        print(e)
        sys.exit(0)                      # the reference prints and exits 0 (main.py:44-46, 70-72)
    if stage.get('train') or stage.get('eval') or stage.get('run'):
        model.build()
    if stage.get('train'):
        model.train()
    if stage.get('eval'):
        model.eval()
    if stage.get('run'):
        model.run()


if __name__ == '__main__':
    main()
