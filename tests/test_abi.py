"""CPU-side checks: the C-ABI library loads and exports every symbol include/vpho_hip.h declares; host-side packing."""
import ctypes
import os
import re

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'vpho_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(vpho_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(os.path.join(ROOT, 'vpho_amd', 'libvpho_hip.so'))
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/vpho_hip.h but not exported'
    lib.vpho_abi_version.restype = ctypes.c_int
    assert lib.vpho_abi_version() == 12


def test_library_exports_nothing_but_the_header():
    """-fvisibility=hidden + VPHO_API + the linker version script (vpho_amd/build.py): the dynamic symbol table IS the C ABI --
    no C++ helper (vpho::fail, prof_record), kernel host stub, __hip_cuid_* or weak STL instantiation leaks out of the library"""
    import subprocess
    import __graft_entry__ as g
    g.build()
    out = subprocess.run(['nm', '-D', '--defined-only', os.path.join(ROOT, 'vpho_amd', 'libvpho_hip.so')], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if l.strip())
    assert exported == _declared(), (sorted(set(exported) - set(_declared())), sorted(set(_declared()) - set(exported)))
    hdr = open(os.path.join(ROOT, 'include', 'vpho_hip.h')).read()
    assert hdr.count('VPHO_API ') - 1 == len(exported)            # every declaration carries the export attribute (+ the #define)


def test_product_library_carries_no_ablation_or_stamp_code():
    """Switches that make a kernel produce WRONG results (timing ablations: skipped loads / barriers / stores) and the in-kernel clock
    stamps exist only as compile-time macros of diagnostic builds (scripts/kernel_ablate.sh, scripts/build_stamps.sh): the product
    library reads no such environment variable, exports no diagnostic entry point, and its build defines none of the macros
    (VPHO_CONV_DBG bits 1 / 2 / 4 and VPHO_WINO_ABL used to be one stray environment variable away)."""
    import subprocess
    import __graft_entry__ as g
    from vpho_amd import build as B
    g.build()
    lib = os.path.join(ROOT, 'vpho_amd', 'libvpho_hip.so')
    strs = subprocess.run(['strings', '-n', '6', lib], capture_output=True, text=True, check=True).stdout
    env = sorted(set(re.findall(r'VPHO_[A-Z0-9_]+', strs)))
    assert 'VPHO_WINO_ABL' not in env and not [e for e in env if 'ABL' in e or 'STAMP' in e], env
    assert 'vpho_diag_' not in strs
    macros = ('CONV_ABLATE', 'WINO_ABLATE', 'WINO8_ABLATE', 'FK_ABLATE', 'VPHO_CLOCK_STAMPS')
    assert not [f for f in B.FLAGS if any(m in f for m in macros)], B.FLAGS
    src = open(os.path.join(ROOT, 'vpho_amd', 'csrc', 'conv_igemm.hip')).read()
    # the run-time VPHO_CONV_DBG keeps only its two bit-identical A/B orders
    assert re.search(r'g\.dbg\s*=\s*dbg_env \? \(atoi\(dbg_env\) & \(8 \| 16\)\)', src), 'VPHO_CONV_DBG must be masked to its bit-identical bits'
    assert not re.search(r'g\.dbg & (1|2|4)\b', src)


def test_state_dict_layout_matches_reference_contract(model_cpu):
    sd = model_cpu.state_dict()
    assert len([k for k in sd if k.startswith('feature_extractor.')]) == 530
    assert len([k for k in sd if k.startswith('encoder_hand.')]) == 170
    assert sd['denoiser_hand.head.head.0.weight'].shape == (32, 1408, 256)
    assert sd['denoiser_obj.head.head.2.weight'].shape == (3, 256, 3)
    assert sd['head_hm_hand.deconv_layers.0.weight'].shape == (128, 64, 4, 4)
    assert sd['cross_hand.pose_embedder.pe'].shape == (5000, 1, 512)
    assert sd['cross_obj.attn.layers.0.self_attn.in_proj_weight'].shape == (1536, 512)
    assert sd['head_physics.anchor'].shape == (8, 3)
    assert 'head_obj.point_002_master_chef_can' in sd and 'head_mano.mano_layer.th_posedirs' in sd


def test_deconv_phase_packing_is_equivalent_on_cpu():
    """pack_deconv4x4s2 (host logic) reproduces ConvTranspose2d(k4,s2,p1) with four stride-1 2x2 convolutions."""
    from vpho_amd.model.pack import pack_deconv4x4s2
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(1, 6, 5, 5, generator=g), torch.randn(6, 4, 4, 4, generator=g)
    ref = F.conv_transpose2d(x, w, None, 2, 1)
    out = torch.zeros_like(ref)
    for (py, px), (wp, pady, padx) in pack_deconv4x4s2(w).items():
        wk = wp.view(4, 2, 2, 6).permute(0, 3, 1, 2)
        xp = F.pad(x, (padx, 1 - padx, pady, 1 - pady))
        out[:, :, py::2, px::2] = F.conv2d(xp, wk)
    assert torch.allclose(out, ref, atol=1e-5)


def test_bn_folding_on_cpu(sd):
    from vpho_amd.model.pack import fold_conv_bn
    p = 'feature_extractor.layer1_h.0.0'
    w, b = fold_conv_bn(sd, p + '.conv1', p + '.bn1')
    x = torch.randn(2, 64, 5, 5)
    ref = F.batch_norm(F.conv2d(x, sd[p + '.conv1.weight']), sd[p + '.bn1.running_mean'], sd[p + '.bn1.running_var'],
                       sd[p + '.bn1.weight'], sd[p + '.bn1.bias'], False, 0.0, 1e-5)
    got = F.conv2d(x, w.view(64, 1, 1, 64).permute(0, 3, 1, 2), b)
    assert torch.allclose(got, ref, atol=1e-5)


def _header_structs():
    """{struct name: [(field name, 'ptr' | 'int' | 'float' | 'double' | 'longlong'), ...]} from include/vpho_hip.h"""
    import re
    txt = open(os.path.join(ROOT, 'include', 'vpho_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', ' ', txt, flags=re.S)
    out = {}
    for body, name in re.findall(r'typedef\s+struct(?:\s+\w+)?\s*\{(.*?)\}\s*(\w+)\s*;', txt, flags=re.S):
        fields = []
        for decl in body.split(';'):
            decl = ' '.join(decl.split())
            if not decl:
                continue
            m = re.match(r'^(const\s+)?(unsigned\s+char|long\s+long|\w+)\s*(.*)$', decl)
            base, rest = m.group(2), m.group(3)
            for item in rest.split(','):
                item = item.strip()
                ptr = item.startswith('*') or base.endswith('*')
                fields.append((item.lstrip('* '), 'ptr' if ptr else {'int': 'int', 'float': 'float', 'double': 'double', 'long long': 'longlong'}[base]))
        out[name] = fields
    return out


def test_ctypes_structures_mirror_the_header_field_for_field():
    """A binding whose Structure is one field short makes the C side read past it (found in INTEGRATION.md's example): names, order and
    kind of every field of every struct the ABI passes by pointer, header against vpho_amd/ops.py."""
    from vpho_amd import ops
    hs = _header_structs()
    pairs = {'vpho_conv_desc': ops.ConvDesc, 'vpho_score_weights': ops.ScoreWeights, 'vpho_ode_stats': ops.OdeStats,
             'vpho_mano_tables': ops.ManoTables, 'vpho_obj_tables': ops.ObjTables, 'vpho_anchor_tables': ops.AnchorTables,
             'vpho_obj_metric_tables': ops.ObjMetricTables}
    kind = {ctypes.c_void_p: 'ptr', ctypes.c_int: 'int', ctypes.c_float: 'float', ctypes.c_double: 'double', ctypes.c_longlong: 'longlong'}
    for name, cls in pairs.items():
        assert name in hs, name
        got = [(f[0], kind[f[1]]) for f in cls._fields_]
        assert got == hs[name], f'{name}: header {hs[name]} vs ctypes {got}'
    # the example a maintainer would copy out of INTEGRATION.md names the same fields of the sampler's weight struct
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    for f, _ in hs['vpho_score_weights']:
        assert f"'{f}'" in doc, f'INTEGRATION.md example lacks vpho_score_weights.{f}'


def _header_prototypes():
    """{function: [kind of every parameter]} with kind in ptr / int / float / double / longlong"""
    import re
    txt = open(os.path.join(ROOT, 'include', 'vpho_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', ' ', txt, flags=re.S)
    txt = re.sub(r'typedef\s+struct(?:\s+\w+)?\s*\{.*?\}\s*\w+\s*;', ' ', txt, flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r'\b(int|long long|const char\s*\*|void)\s+(vpho_\w+)\s*\(([^)]*)\)\s*;', txt):
        kinds = []
        for a in args.split(','):
            a = ' '.join(a.split())
            if not a or a == 'void':
                continue
            if '*' in a:
                kinds.append('ptr')
            elif a.startswith('long long'):
                kinds.append('longlong')
            else:
                kinds.append({'int': 'int', 'float': 'float', 'double': 'double', 'unsigned': 'int'}[a.replace('const ', '').split()[0]])
        protos[name] = kinds
    return protos


def test_every_ctypes_call_site_passes_what_the_header_declares():
    """Static check of vpho_amd/ops.py against include/vpho_hip.h: each `_call('vpho_x', ...)` passes as many arguments as the prototype
    has before its trailing stream, and every argument built by a typed wrapper (I / F / LL / c_double / tensor pointer / byref) has the
    declared kind -- a swapped or missing argument would otherwise only show on the GPU, as a wrong result."""
    import ast
    protos = _header_prototypes()
    assert len(protos) >= 80
    src = open(os.path.join(ROOT, 'vpho_amd', 'ops.py')).read()
    wrap = {'I': 'int', 'F': 'float', 'LL': 'longlong', '_f32': 'ptr', '_f64': 'ptr', '_i32': 'ptr', '_u8': 'ptr', '_ptr': 'ptr', '_at': 'ptr'}

    def kind(node):
        if isinstance(node, ast.Call):
            f = node.func
            if isinstance(f, ast.Name) and f.id in wrap:
                return wrap[f.id]
            if isinstance(f, ast.Attribute) and f.attr in ('byref', 'c_void_p'):
                return 'ptr'
            if isinstance(f, ast.Attribute) and f.attr == 'c_double':
                return 'double'
            if isinstance(f, ast.Attribute) and f.attr == 'c_longlong':
                return 'longlong'
        if isinstance(node, ast.Constant) and node.value is None:
            return 'ptr'
        return None                      # built elsewhere (a ctypes array, a local): not judged

    seen, judged = set(), 0
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == '_call':
            name = node.args[0].value
            assert name in protos, f'{name} is called but not declared in include/vpho_hip.h'
            want = protos[name]
            assert want and want[-1] == 'ptr', f'{name}: the last parameter must be the stream'
            got = node.args[1:]
            if any(isinstance(a, ast.Starred) for a in got):
                continue
            assert len(got) == len(want) - 1, f'{name}: {len(got)} arguments at line {node.lineno}, header has {len(want) - 1} before the stream'
            for i, (a, w) in enumerate(zip(got, want)):
                k = kind(a)
                if k is not None:
                    judged += 1
                    assert k == w, f'{name} argument {i} at line {node.lineno}: passes {k}, header declares {w}'
            seen.add(name)
    assert len(seen) >= 60 and judged >= 600, (len(seen), judged)
    # entry points called directly (not through _call) declare argtypes: those must spell the prototype
    from vpho_amd import ops
    kinds = {ctypes.c_void_p: 'ptr', ctypes.c_int: 'int', ctypes.c_float: 'float', ctypes.c_double: 'double', ctypes.c_longlong: 'longlong'}
    typed = 0
    for name, want in protos.items():
        at = getattr(ops.lib, name).argtypes
        if at is None:
            continue
        got = [kinds.get(t, 'ptr' if hasattr(t, 'contents') or hasattr(t, '_type_') and isinstance(t._type_, type) else None) for t in at]
        assert got == want, f'{name}.argtypes {got} vs header {want}'
        typed += 1
    assert typed >= 4
    for node in ast.walk(ast.parse(src)):                 # ... and every direct call has argtypes behind it or typed wrappers only
        if (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and isinstance(node.func.value, ast.Name)
                and node.func.value.id == 'lib' and node.func.attr.startswith('vpho_') and node.func.attr in protos):
            name = node.func.attr
            if getattr(ops.lib, name).argtypes is None:
                assert len(node.args) == len(protos[name]), f'{name}: {len(node.args)} arguments at line {node.lineno}, header has {len(protos[name])}'
                for i, (a, w) in enumerate(zip(node.args, protos[name])):
                    k = kind(a) or ('ptr' if isinstance(a, ast.Call) and isinstance(a.func, ast.Name) and a.func.id == '_stream' else None)
                    assert k == w, f'{name} argument {i} at line {node.lineno}: passes {k} without argtypes, header declares {w}'
