"""Per-tensor report of the end-to-end training-step gradients vs the reference fixture (GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from tests.test_gpu_train_step import load_case, compare_gradients
from vpho_amd.train_step import DiffusionTrainStep
sd, data, draws, G = load_case()
step = DiffusionTrainStep(sd, 'cuda', loss_weights=dict(hm_hand=1e3, hm_obj=1e3))
gt_h, gt_o = torch.from_numpy(G['gt_hand6d']).cuda(), torch.from_numpy(G['gt_obj']).cuda()
L, grads = step.loss_and_grads(data, gt_h, gt_o, draws)
for k in ('diff_hand_loss', 'diff_obj_loss', 'hm_hand_loss', 'hm_obj_loss'):
    print(k, float(L[k]), float(G[k]))
rep = []
bad, ours, theirs = compare_gradients(G, grads, rep)
print('median deviation from fp64: ours %.3e, reference fp32 %.3e' % (ours, theirs))
rep.sort(key=lambda r: -r[3])
print('worst by max sampled error / rms:')
for r in rep[:25]:
    print('  %-60s norm rel %.2e  med %.3e  max %.3e  noise %.3e %s' % r)
rep.sort(key=lambda r: -r[1])
print('worst by norm:')
for r in rep[:15]:
    print('  %-60s norm rel %.2e  med %.3e  max %.3e  noise %.3e %s' % r)
print('bad', len(bad), 'of', len(rep))
torch.cuda.synchronize(); t = time.time()
for _ in range(3):
    step.step(data, gt_h, gt_o, draws)
torch.cuda.synchronize(); print('ms/step (bs=12, reps=2)', (time.time() - t) / 3 * 1e3)
for b in bad:
    print('BAD %-60s norm rel %.2e med %.3e max %.3e noise %.3e' % b)
