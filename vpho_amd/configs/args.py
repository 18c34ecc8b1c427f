"""Run configuration: the module-global ``cfg`` the model reads at call time.

Mirrors the behaviour of the reference's lib/configs/args.py:4-260 for the flags the inference path and the
``main.py`` entry point use (same names, same defaults, same "argparse at import" semantics), except that unknown
flags are tolerated (``parse_known_args``) so that importing the package under pytest / torchrun does not abort.
"""
import argparse
import sys


class Config:
    def __init__(self):
        self.mode = 'train'
        self.eval_full = False
        self.mark = ''
        self.random_seed = 0
        self.output_dir = 'output'
        self.checkpoint = ''
        self.pretrain = ''
        self.remove_pretrained_keys = []
        self.dataset_name = 'dexycb'
        self.patch_size = 256
        self.batch_size = 64
        self.eval_batch_size = 32
        self.num_batches = 4          # synthetic evaluation only (no dataset in this build)
        self.model = 'vpho_net'
        self.sde_mode = 've'
        self.repeat_num = 20
        self.sampler = 'ode'
        self.sampling_steps = 500
        self.heatmap_size = 64
        self.roi_size = 32
        self.sample_T0 = 0.65
        self.sample_num = 50
        self.topk_hand = 15
        self.topk_obj = 5
        self.asset_root = 'asset'
        self.base_learning_rate = 2e-4
        self.gradient_clip = -1.0
        self.train_scope = 'full'     # 'full': backbone + heads + encoders + score networks; 'score': score networks on frozen features
        self.weight_diff_hand_loss = 1.0
        self.weight_diff_obj_loss = 1.0
        self.weight_hm_hand_loss = 1.0
        self.weight_hm_obj_loss = 1.0
        self.weight_vert_loss = 1.0
        self.weight_joint_loss = 1.0
        self.weight_mano_pose_loss = 1.0
        self.weight_mano_shape_loss = 1.0
        self.weight_force_loss = 1.0
        self.weight_gravity_loss = 1.0
        self.weight_torque_loss = 1.0
        self.weight_supervised_loss = 1.0
        self.weight_CoM_loss = 1.0
        self.cross_dropout = 0.1      # the reference's nn.TransformerEncoderLayer / PositionalEncoding default (cross_module.py:64,104-107)


def _parser():
    p = argparse.ArgumentParser(description='Hand-Object Pose Estimation (MI355X hot path)')
    p.add_argument('--mode', type=str, default='train', choices=['train', 'eval', 'infer'])
    p.add_argument('--eval_full', action='store_true')
    p.add_argument('--mark', type=str, default='')
    p.add_argument('--random_seed', type=int, default=0)
    p.add_argument('--output_dir', type=str, default='output')
    p.add_argument('--checkpoint', type=str, default='')
    p.add_argument('--pretrain', type=str, default='')
    p.add_argument('--remove_pretrained_keys', nargs='+', default=[])
    p.add_argument('--dataset_name', type=str, default='dexycb', choices=['dexycb', 'ho3d'])
    p.add_argument('--patch_size', type=int, default=256)
    p.add_argument('--batch_size', type=int, default=64)
    p.add_argument('--eval_batch_size', type=int, default=32)
    p.add_argument('--num_batches', type=int, default=4)
    p.add_argument('--model', type=str, default='vpho_net', choices=['vpho_net'])
    p.add_argument('--sde_mode', type=str, choices=['ve'], default='ve')
    p.add_argument('--repeat_num', type=int, default=20)
    p.add_argument('--sampler', type=str, choices=['ode'], default='ode')
    p.add_argument('--sampling_steps', type=int, default=500)
    p.add_argument('--heatmap_size', type=int, default=64)
    p.add_argument('--roi_size', type=int, default=32)
    p.add_argument('--sample_T0', type=float, default=0.65)
    p.add_argument('--sample_num', type=int, default=50)
    p.add_argument('--topk_hand', type=int, default=15)
    p.add_argument('--topk_obj', type=int, default=5)
    p.add_argument('--asset_root', type=str, default='asset')
    p.add_argument('--base_learning_rate', type=float, default=2e-4)
    p.add_argument('--gradient_clip', type=float, default=-1.)
    p.add_argument('--train_scope', type=str, default='full', choices=['full', 'score'])
    p.add_argument('--weight_diff_hand_loss', type=float, default=1.0)
    p.add_argument('--weight_diff_obj_loss', type=float, default=1.0)
    p.add_argument('--weight_hm_hand_loss', type=float, default=1e3)
    p.add_argument('--weight_hm_obj_loss', type=float, default=1e3)
    p.add_argument('--weight_vert_loss', type=float, default=1e4)
    p.add_argument('--weight_joint_loss', type=float, default=1e4)
    p.add_argument('--weight_mano_pose_loss', type=float, default=10)
    p.add_argument('--weight_mano_shape_loss', type=float, default=1.0)
    p.add_argument('--weight_force_loss', type=float, default=1.0)
    p.add_argument('--weight_gravity_loss', type=float, default=1.0)
    p.add_argument('--weight_torque_loss', type=float, default=30.0)
    p.add_argument('--weight_supervised_loss', type=float, default=10)
    p.add_argument('--weight_CoM_loss', type=float, default=1e2)
    p.add_argument('--cross_dropout', type=float, default=0.1)
    return p


cfg = Config()
_args, _unknown = _parser().parse_known_args(sys.argv[1:])
for _k, _v in vars(_args).items():
    if hasattr(cfg, _k):
        setattr(cfg, _k, _v)
    else:
        raise ValueError(f"Invalid config key: {_k}")
