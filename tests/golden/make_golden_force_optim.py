"""Golden vectors of the reference's OWN pseudo-force optimisation loop: ``ForceOptimizer.optimize_batch``
(lib/engine/force_optimization.py:110-207) is executed unchanged -- all 3000 AdamW iterations, 300 of them in phase 1 -- on one
synthetic batch, with the reference's ``HeadForce`` (lib/model/physics.py), ``VERT2ANCHOR`` (lib/utils/physics_fn.py) and
``torch.optim.AdamW``.  Snapshots of the optimised parameters are taken inside ``accel.backward`` after 40 / 400 / 1000 / 3000
steps; the labels the loop hands to ``save_force`` are stored as well.

What is stubbed (the module is not importable as shipped): ``lib.dataset.dexycb4`` / ``lib.dataset.ho3d2`` (absent
from the tree; only the names DexYCBDataset / YCB_MESHES / HO3DDataset_* are imported), cv2 / natsort (imported by
lib.utils.misc_fn, unused here), pytorch3d (bound to the oracle's restatement as in make_golden.py), and the object is created
without BaseTrainer.__init__ (no accelerate process group, no logger / dataloader): ``accel`` is a three-method stand-in
(autocast, backward, num_processes = 1).  Run in the build container only (needs /root/reference)."""
import contextlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

B = 8
SNAP = (40, 400, 1000, 3000)


def inputs(assets, seed=1):
    """The same synthetic pairs tests/test_gpu_force_optim.py::_inputs builds."""
    g = torch.Generator().manual_seed(seed)
    v = torch.as_tensor(assets['mano']['v_template'])[None] + torch.randn(B, 778, 3, generator=g) * 0.002 + torch.tensor([0.0, 0.0, 0.7])
    grav = torch.nn.functional.normalize(torch.randn(B, 1, 3, generator=g), dim=-1)
    com = torch.tensor([0.05, 0.0, 0.7]) + torch.randn(B, 1, 3, generator=g) * 0.02
    fc = torch.rand(B, 32, generator=g)
    grasped = torch.rand(B, generator=g) < 0.8
    return v.contiguous(), grav.contiguous(), com.contiguous(), fc.contiguous(), grasped


def main():
    from vpho_amd.assets import synthetic_assets, YCB_NAMES
    assets = synthetic_assets(0)
    tmp = tempfile.mkdtemp(prefix='vpho_golden_fo_')
    MG.write_assets(tmp, assets)
    os.chdir(tmp)
    sys.argv = ['force_optim.py']
    sys.path.insert(0, MG.REF)
    MG.install_stubs(assets)
    ycb = sys.modules['lib.dataset.base'].YCB_MESHES
    MG._stub('lib.dataset.dexycb4', DexYCBDataset=object, YCB_MESHES=ycb)
    MG._stub('lib.dataset.ho3d2', HO3DDataset_Train=object, HO3DDataset_Test=object)
    MG._stub('cv2')
    MG._stub('natsort', natsorted=sorted)
    import lib.engine.force_optimization as FO

    v, grav, com, fc, grasped = inputs(assets)
    opt = object.__new__(FO.ForceOptimizer)                    # no BaseTrainer.__init__: no Accelerator / logger / dataloaders
    opt.cfg = types.SimpleNamespace(batch_size=B, eval_batch_size=B)
    opt.device = torch.device('cpu')
    opt.num_anchor = 32
    opt.get_model()                                            # HeadForce(1) -- the reference's module
    opt.get_optimizer()                                        # scale / weight parameters + the two AdamW optimisers
    opt.get_obj_mesh()
    snaps = {}
    calls = [0]

    class Accel:
        num_processes = 1
        is_local_main_process = False

        def autocast(self):
            return contextlib.nullcontext()

        def backward(self, loss):                              # called once per iteration, BEFORE that iteration's step
            if calls[0] in SNAP:
                snaps[calls[0]] = (opt.scale.detach().clone(), opt.weight.detach().clone())
            calls[0] += 1
            loss.backward()

    opt.accel = Accel()
    rec = {}
    opt.save_viz = lambda **kw: rec.update(force_point=kw['force_point'].detach().clone(), force_global_viz=kw['force_global'].detach().clone())
    opt.save_force = lambda **kw: rec.update(force_local=kw['force_local'].detach().clone(), force_global=kw['force_global'].detach().clone())
    batch = dict(force_contact=fc.clone(), gt_hand_vert_flip=v.clone(), gravity=grav.clone(), obj_CoM=com.clone(),
                 obj_id=torch.zeros(B, dtype=torch.long), obj_name=[YCB_NAMES[0]] * B,
                 gt_obj=torch.cat([torch.tensor([1.0, 0, 0, 0, 1, 0]).repeat(B, 1), torch.zeros(B, 3)], 1),
                 is_grasped=grasped.clone(), rgb_path=[f'img_{i}.jpg' for i in range(B)],
                 is_right=torch.ones(B, dtype=torch.bool))       # right hands: inputs are already in the flipped frame
    opt.training_dataloader = [batch]
    opt.optimize_batch()
    assert calls[0] == 3000
    snaps[3000] = (opt.scale.detach().clone(), opt.weight.detach().clone())
    P = dict(B=np.array(B), seed=np.array(1), force_local=rec['force_local'].numpy(), force_global=rec['force_global'].numpy(),
             force_point=rec['force_point'].numpy())
    for k, (s, w) in snaps.items():
        P[f'scale_{k}'], P[f'weight_{k}'] = s.numpy(), w.numpy()
    path = os.path.join(HERE, 'golden_force_optim.npz')
    np.savez_compressed(path, **P)
    print({k: v.shape for k, v in P.items()}, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
