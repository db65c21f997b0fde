"""Flag system compatible with the reference's arguments.py:5-177 (same flag names, types,
defaults and the same store_false gotchas: --not_pool_vis, --use_spec), table-driven.

``ArgParser().parse_train_arguments(argv)`` returns the Namespace the step code reads;
``train_music_args()`` is the flag set of the config of record, scripts/train_MUSIC.sh.
"""
import argparse

# (flag, kwargs) — model / data / misc flags (arguments.py:5-95)
_BASE = [
    ("--id", dict(default="")), ("--num_mix", dict(default=2, type=int)),
    ("--arch_sound", dict(default="unet7")), ("--arch_frame", dict(default="resnet18dilated")),
    ("--arch_synthesizer", dict(default="linear")), ("--fusion_type", dict(default="con")),
    ("--weights_sound", dict(default="")), ("--weights_frame", dict(default="")),
    ("--weights_synthesizer", dict(default="")), ("--num_channels", dict(default=32, type=int)),
    ("--num_frames", dict(default=1, type=int)), ("--stride_frames", dict(default=1, type=int)),
    ("--img_pool", dict(default="maxpool")), ("--img_activation", dict(default="sigmoid")),
    ("--sound_activation", dict(default="no")), ("--output_activation", dict(default="sigmoid")),
    ("--binary_mask", dict(default=1, type=int)), ("--mask_thres", dict(default=0.5, type=float)),
    ("--loss", dict(default="l1")), ("--weighted_loss", dict(default=0, type=int)),
    ("--log_freq", dict(default=1, type=int)), ("--vis_channels", dict(default=512, type=int)),
    ("--not_pool_vis", dict(action="store_false", default=True)),
    ("--num_gpus", dict(default=1, type=int)), ("--batch_size_per_gpu", dict(default=32, type=int)),
    ("--workers", dict(default=32, type=int)), ("--num_val", dict(default=-1, type=int)),
    ("--num_vis", dict(default=40, type=int)), ("--audLen", dict(default=65535, type=int)),
    ("--audRate", dict(default=11025, type=int)), ("--stft_frame", dict(default=1022, type=int)),
    ("--stft_hop", dict(default=256, type=int)), ("--imgSize", dict(default=224, type=int)),
    ("--frameRate", dict(default=8, type=float)), ("--load_clips", dict(action="store_true", default=False)),
    ("--clip_len", dict(default=32, type=int)), ("--seed", dict(default=1234, type=int)),
    ("--ckpt", dict(default="./ckpt")), ("--disp_iter", dict(type=int, default=20)),
    ("--eval_epoch", dict(type=int, default=1)),
]
# train flags (arguments.py:97-136)
_TRAIN = [
    ("--mode", dict(default="train")), ("--list_train", dict(nargs="+", default=["data/train.csv"])),
    ("--list_val", dict(nargs="+", default=["data/val.csv"])),
    ("--av_list_train", dict(nargs="+", default=["data/train_av.csv"])),
    ("--ao_list_train", dict(nargs="+", default=["data/train_ao.csv"])),
    ("--num_epoch", dict(default=100, type=int)), ("--num_iters", dict(default=120000, type=int)),
    ("--eval_iter", dict(default=7500, type=int)), ("--iter_per_av", dict(default=2, type=int)),
    ("--lr_frame", dict(default=1e-4, type=float)), ("--lr_sound", dict(default=1e-3, type=float)),
    ("--lr_motion", dict(default=1e-4, type=float)), ("--lr_synthesizer", dict(default=1e-3, type=float)),
    ("--lr_steps", dict(nargs="+", type=int, default=[20000, 40000])),
    ("--start_av_first", dict(action="store_true", default=False)), ("--num_fsteps", dict(default=40000, type=int)),
    ("--beta1", dict(default=0.9, type=float)), ("--weight_decay", dict(default=1e-4, type=float)),
    ("--train_repeat", dict(default=100, type=int)),
]
# other flags (arguments.py:143-169); --load_ckpt is type=str with default False, as in the reference
_OTHER = [
    ("--load_ckpt", dict(type=str, default=False)), ("--use_spec", dict(action="store_false", default=True)),
    ("--rate_dc", dict(type=float, default=1.0)), ("--rate_sc", dict(type=float, default=0.05)),
    ("--rate_sv", dict(type=float, default=0.0)), ("--margin", dict(type=float, default=3.0)),
    ("--max_silent", dict(type=float, default=0.67)), ("--val_repeat", dict(type=int, default=12)),
    ("--match_weight", dict(default=0.6, type=float)), ("--one_frame", dict(action="store_true", default=False)),
    ("--fix_vis", dict(action="store_true", default=False)), ("--att_type", dict(type=str, default="cos")),
    # SoP++/main.py:674-679 reads args.train_steps (three stage boundaries) but no parser of the reference defines it;
    # given, train.py runs the SoP++ variant (basis U-Net + attention module + synthesizer) on the 3-stage schedule
    ("--train_steps", dict(nargs=3, type=int, default=None)),
]


class ArgParser(object):
    def __init__(self):
        self.parser = argparse.ArgumentParser()
        self._add(_BASE)

    def _add(self, table):
        for flag, kw in table:
            self.parser.add_argument(flag, **kw)

    def add_train_arguments(self):
        self._add(_TRAIN)

    def add_other_arguments(self):
        self._add(_OTHER)

    def print_arguments(self, args):
        print("Input arguments:")
        for key, val in vars(args).items():
            print("{:16} {}".format(key, val))

    def parse_train_arguments(self, argv=None, verbose=True):
        self.add_train_arguments()
        self.add_other_arguments()
        args = self.parser.parse_args(argv)
        if verbose:
            self.print_arguments(args)
        return args


TRAIN_MUSIC_FLAGS = (
    "--id Exp5_BaseSig --av_list_train data/train.csv --ao_list_train data/train.csv --list_val data/val.csv "
    "--start_av_first --num_fsteps 0 --arch_sound unet7 --arch_synthesizer linear --arch_frame resnet18dilated "
    "--img_pool maxpool --num_channels 2 --img_activation relu --output_activation sigmoid --vis_channels 256 "
    "--fusion_type hidsep --not_pool_vis --att_type sig --binary_mask 1 --loss bce --weighted_loss 1 --num_mix 2 "
    "--log_freq 1 --num_frames 3 --stride_frames 8 --frameRate 30 --audLen 65535 --audRate 11025 --num_gpus 2 "
    "--workers 4 --batch_size_per_gpu 16 --lr_frame 1e-4 --lr_sound 1e-3 --lr_synthesizer 1e-3 "
    "--lr_steps 50000 70000 90000 --num_iters 95001 --iter_per_av 2 --eval_iter 1000 --train_repeat 50 "
    "--disp_iter 20 --num_vis 100 --num_val 256 --rate_dc 1 --max_silent 0.87 --mask_thres 0.5 "
    "--match_weight 0.1 --one_frame").split()


def train_music_args(extra=()):
    """The Namespace produced by scripts/train_MUSIC.sh (config of record), plus overrides."""
    return ArgParser().parse_train_arguments(list(TRAIN_MUSIC_FLAGS) + list(extra), verbose=False)
