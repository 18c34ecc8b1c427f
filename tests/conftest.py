import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]          # vpho_amd.configs.args parses sys.argv at import, like the reference's lib/configs/args.py
# libgomp reads this when torch loads it.  With the default policy the OpenMP workers spin between parallel regions; on a small VM the
# scheduler can leave two spinning threads on one vCPU for minutes (another vCPU idle), and every barrier of the CPU oracle then
# costs a time slice: the suite went from 80 s to 220 s that way.  Sleeping workers are woken onto idle CPUs instead.
os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the CPU oracle: one OpenMP worker per CPU this process may really use (cgroup quota), not per hardware thread of the host
    import torch
    from vpho_amd.hostcpu import usable_cpus
    torch.set_num_threads(min(torch.get_num_threads(), usable_cpus()))


@pytest.fixture(scope='session')
def assets():
    from vpho_amd.assets import synthetic_assets
    return synthetic_assets(0)


@pytest.fixture(scope='session')
def model_cpu(assets):
    from vpho_amd.model.VPHO import vpho_net
    from vpho_amd.synth import synth_state_dict
    m = vpho_net(assets)
    m.load_state_dict(synth_state_dict(m, seed=1))
    return m.eval()


@pytest.fixture(scope='session')
def sd(model_cpu):
    return {k: v.clone() for k, v in model_cpu.state_dict().items()}


@pytest.fixture(scope='session')
def model_contrast_cpu(assets):
    """vpho_amd.synth.bench_state_dict: seeded weights with high-contrast heat-maps and conditioned score networks -- the weights
    of bench.py, of the README-config reference fixture and of the README-size parity tests."""
    from vpho_amd.model.VPHO import vpho_net
    from vpho_amd.synth import bench_state_dict
    m = vpho_net(assets)
    m.load_state_dict(bench_state_dict(m, seed=1))
    return m.eval()


@pytest.fixture(scope='session')
def sd_contrast(model_contrast_cpu):
    return {k: v.clone() for k, v in model_contrast_cpu.state_dict().items()}
