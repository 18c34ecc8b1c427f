"""Golden vectors for the object evaluation metrics by calling the reference's own ``TesterObject`` (lib/engine/test.py).

Run in the build container only.  Same stubs as make_golden.py; in addition
* ``TesterObject.__init__`` opens asset/2023_NIPS_DeepSimHO/assets_models_info.json (absent; only used by the symmetric
  corner error, which ``__call__`` has commented out) -> the instance is created with ``object.__new__`` and given the
  ``obj_mesh`` table only;
* the criteria move tensors with ``.cuda()``; there is no GPU here -> ``torch.Tensor.cuda`` is an identity for this script
  (same fp32 arithmetic on the CPU).
Writes golden_objmetrics.npz: inputs (pd_rt, gt_rt, cam_intr, object ids) and the per-sample metric table.
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def main():
    from vpho_amd.assets import synthetic_assets
    from oracle.metrics import OBJ_METRIC_NAMES
    assets = synthetic_assets(0)
    tmp = tempfile.mkdtemp(prefix='vpho_golden_obj_')
    MG.write_assets(tmp, assets)
    os.chdir(tmp)
    sys.argv = ['main.py', '--mode', 'eval']
    sys.path.insert(0, MG.REF)
    MG.install_stubs(assets)
    ycb = sys.modules['lib.dataset.base'].YCB_MESHES
    for k, v in assets['ycb'].items():          # fp64 tables like trimesh's (base.py:222-244)
        ycb[k] = {kk: (np.asarray(vv, np.float64) if isinstance(vv, np.ndarray) else vv) for kk, vv in ycb[k].items()}
        ycb[k]['bbox3d'] = np.asarray(v['bbox3d'], np.float64)
        ycb[k]['verts'] = np.asarray(v['verts'], np.float64)
        ycb[k]['verts_sampled'] = np.asarray(v['verts_sampled'], np.float64)
        ycb[k]['diameter'] = v['diameter']
    torch.Tensor.cuda = lambda self, *a, **kw: self
    from lib.engine.test import TesterObject
    tester = object.__new__(TesterObject)
    tester.obj_mesh = ycb

    rng = np.random.default_rng(77)
    names = list(ycb.keys())
    n = 10
    from oracle import rotations as R
    def rand_rt(scale_rot, scale_t, base=None):
        aa = torch.from_numpy(rng.normal(size=(n, 3)) * scale_rot)
        Rm = R.axis_angle_to_matrix(aa).numpy()
        t = rng.normal(size=(n, 3)) * scale_t
        if base is None:
            t = t + np.array([0.0, 0.0, 0.7])
            return np.concatenate([Rm, t[:, :, None]], -1)
        return np.concatenate([Rm @ base[:, :, :3], (base[:, :, 3] + t)[:, :, None]], -1)
    gt_rt = rand_rt(1.0, 0.05).astype(np.float32)
    # predictions from nearly exact to far off, so that every threshold metric sees both outcomes
    pd_rt = rand_rt(np.linspace(0.002, 0.6, n)[:, None], np.linspace(0.0005, 0.04, n)[:, None], gt_rt.astype(np.float64)).astype(np.float32)
    f = rng.uniform(400, 600, size=n)
    cam = np.stack([np.array([[fi, 0, 128.0], [0, fi, 128.0], [0, 0, 1.0]]) for fi in f]).astype(np.float32)
    obj_idx = rng.integers(0, len(names), size=n)
    obj_name = np.array([names[i] for i in obj_idx])
    res = tester({'pd_rt': pd_rt, 'gt_rt': gt_rt, 'obj_name': obj_name, 'cam_intr': cam})
    # REP5 is computed by __call__ (cal_REP5, test.py:518-519) but not put into its result dict: derived here from REP
    col = lambda k: (np.asarray(res['REP']['average_instance']) < 5) if k == 'REP5' else res[k]['average_instance']
    for k in OBJ_METRIC_NAMES:
        print(k, np.asarray(col(k)).shape)
    table = np.stack([np.asarray(col(k), np.float64).reshape(n) for k in OBJ_METRIC_NAMES], -1)
    print(OBJ_METRIC_NAMES)
    print(np.array2string(table, precision=5, suppress_small=True))
    np.savez_compressed(os.path.join(HERE, 'golden_objmetrics.npz'), pd_rt=pd_rt, gt_rt=gt_rt, cam_intr=cam,
                        obj_idx=obj_idx.astype(np.int64), metrics=table)


if __name__ == '__main__':
    main()
