"""Build ``vpho_amd/libvpho_hip.so`` (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

``python -m vpho_amd.build`` or ``build_extension()``; objects are cached under ``vpho_amd/csrc/_obj`` by mtime.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libvpho_hip.so')
ARCH = 'gfx950'
FLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-ffp-contract=off', '-fvisibility=hidden', '-fvisibility-inlines-hidden', '-Wall', '-Wno-unused-function']


def _hipcc():
    for c in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: the vpho_amd HIP extension cannot be built')


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(os.path.dirname(HERE), 'include', 'vpho_hip.h'))
    return max(os.path.getmtime(h) for h in hs)


def build_extension(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    cc = _hipcc()
    hdr = _newest_header()
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src) + '.o')
        objs.append(obj)
        # every kernel's register / LDS / scratch use is kept next to its object (-Rpass-analysis=kernel-resource-usage): an edit that
        # pushes a kernel over its register budget compiles without a word and costs its occupancy -- tests/test_kernel_resources.py reads these
        usage = obj[:-2] + '.usage.txt' if src.endswith('.hip') else None
        stale = not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr) or (usage is not None and not os.path.exists(usage))
        if force or stale:
            cmd = [cc] + FLAGS + (['-x', 'hip', '-Rpass-analysis=kernel-resource-usage'] if src.endswith('.hip') else []) + ['-c', src, '-o', obj]
            jobs.append((cmd, usage))

    def run(job):
        cmd, usage = job if isinstance(job, tuple) else (job, None)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd) + '\n' + r.stdout + r.stderr)
        if usage is not None:
            with open(usage, 'w') as f:
                f.write(''.join(l + '\n' for l in r.stderr.splitlines() if 'kernel-resource-usage' in l))
        elif verbose and r.stderr:
            sys.stderr.write(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        # the C ABI of include/vpho_hip.h is the library's whole dynamic symbol table: -fvisibility=hidden + VPHO_API on the
        # declarations, and a version script for what the compiler exports on its own (kernel host stubs, __hip_cuid_*, weak STL code)
        vmap = os.path.join(OBJ, 'exports.map')
        with open(vmap, 'w') as f:
            f.write('{ global: vpho_*; local: *; };\n')
        run([cc, '-shared', '-fPIC', f'--offload-arch={ARCH}', f'-Wl,--version-script={vmap}', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build_extension(force='--force' in sys.argv, verbose=True))
