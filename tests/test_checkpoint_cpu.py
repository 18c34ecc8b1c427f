"""Checkpoint loading (base_trainer.py:81-96): the reference resumes / evaluates from ``accel.save_state`` directories
(``<save_dir>/checkpoint/epoch_N.state``) and from ``final_model.pt``.  The directory here is written by the REAL
``accelerate.Accelerator.save_state`` (CPU), so the file layout is accelerate's own.  CPU only."""
import os

import pytest
import torch

from vpho_amd.trainer import load_checkpoint_state_dict


@pytest.fixture(scope='module')
def small_module():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Linear(4, 2))


@pytest.mark.parametrize('safe', [True, False])
def test_accelerate_state_directory(tmp_path, small_module, safe):
    from accelerate import Accelerator
    accel = Accelerator(cpu=True)
    model = accel.prepare(small_module)
    d = str(tmp_path / 'checkpoint' / 'epoch_45.state')           # the reference's path pattern (base_trainer.py:26-29,85-89)
    accel.save_state(d, safe_serialization=safe)
    files = sorted(os.listdir(d))
    assert ('model.safetensors' in files) if safe else ('pytorch_model.bin' in files), files
    sd, path = load_checkpoint_state_dict(d)
    assert os.path.dirname(path) == d
    want = small_module.state_dict()
    assert set(sd) == set(want)
    for k in want:
        assert torch.equal(sd[k], want[k]), k


def test_plain_state_dict_file_and_ddp_prefix(tmp_path, small_module):
    p = str(tmp_path / 'final_model.pt')
    torch.save(small_module.state_dict(), p)                       # base_trainer.py:91-96
    sd, path = load_checkpoint_state_dict(p)
    assert path == p and all(torch.equal(sd[k], v) for k, v in small_module.state_dict().items())
    q = str(tmp_path / 'ddp.pt')
    torch.save({'module.' + k: v for k, v in small_module.state_dict().items()}, q)
    sd, _ = load_checkpoint_state_dict(q)
    assert set(sd) == set(small_module.state_dict())


def test_missing_files_raise_a_clear_error(tmp_path):
    with pytest.raises(FileNotFoundError, match='does not exist'):
        load_checkpoint_state_dict(str(tmp_path / 'nope.state'))
    d = tmp_path / 'empty.state'
    d.mkdir()
    (d / 'optimizer.bin').write_bytes(b'x')
    with pytest.raises(FileNotFoundError, match='holds no model file'):
        load_checkpoint_state_dict(str(d))


def test_vpho_net_round_trips_through_an_accelerate_state(tmp_path, model_cpu, assets):
    """All 1079 tensors of vpho_net under the reference's key names survive save_state -> load (strict key equality)."""
    from safetensors.torch import save_file
    from vpho_amd.model.VPHO import vpho_net
    d = tmp_path / 'epoch_1.state'
    d.mkdir()
    sd = {k: v.contiguous().clone() for k, v in model_cpu.state_dict().items()}
    save_file(sd, str(d / 'model.safetensors'))
    got, _ = load_checkpoint_state_dict(str(d))
    m = vpho_net(assets)
    missing, unexpected = m.load_state_dict(got, strict=False)
    assert not missing and not unexpected
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
