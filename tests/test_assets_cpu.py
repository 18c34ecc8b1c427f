"""vpho_amd.assets.load_assets against files in the REFERENCE's on-disk formats (head_mano.py:48-55 -> manopth's MANO_RIGHT.pkl;
physics_fn.py:186-199, hand_fn.py:427-431 -> the CPF anchor files + vert2joint.pkl; dataset/base.py:204-258 -> object_mesh_info.pkl):
round trip bit for bit, loud on damaged or incomplete files, reported when a table falls back to synthetic data."""
import os
import pickle
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))


def _write_reference_layout(root, assets):
    """the files the reference opens, written the way its own loaders expect them"""
    import scipy.sparse as sp
    from make_golden import write_assets                      # the generator of the golden fixtures writes the anchor files for the reference
    write_assets(str(root), assets)                           # -> <root>/asset/ours/vert2joint.pkl, <root>/asset/2021_CVPR_CPF/anchor/*
    a = root / 'asset'
    m = assets['mano']
    (a / 'mano_v1_2' / 'models').mkdir(parents=True)
    with open(a / 'mano_v1_2' / 'models' / 'MANO_RIGHT.pkl', 'wb') as f:      # python-2 era pickle of a dict; J_regressor sparse, like the original
        pickle.dump(dict(v_template=m['v_template'].astype(np.float64), shapedirs=m['shapedirs'].astype(np.float64),
                         posedirs=m['posedirs'].astype(np.float64), J_regressor=sp.csc_matrix(m['J_regressor'].astype(np.float64)),
                         weights=m['weights'].astype(np.float64), f=np.zeros((1538, 3), np.uint32), kintree_table=np.zeros((2, 16), np.int64),
                         hands_components=np.eye(45), hands_mean=np.zeros(45), bs_type='lrotmin', bs_style='lbs'), f, protocol=2)
    from collections import defaultdict
    mesh = defaultdict(dict)
    for k, v in assets['ycb'].items():                        # dataset/base.py:218-256: float64 arrays + keys this build does not read
        mesh[k].update({kk: (np.asarray(vv, np.float64) if kk != 'diameter' else vv) for kk, vv in v.items()})
        mesh[k]['normals_sampled'] = np.zeros((2048, 3))
        mesh[k]['shift'] = np.eye(4)
    with open(a / 'ours' / 'object_mesh_info.pkl', 'wb') as f:
        pickle.dump(mesh, f)
    return str(a)


def test_assets_in_reference_formats_round_trip(tmp_path, assets, capsys):
    from vpho_amd.assets import load_assets
    root = _write_reference_layout(tmp_path, assets)
    got = load_assets(root)
    assert got['synthetic'] is False and all(v != 'synthetic' for v in got['sources'].values()), got['sources']
    for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'weights'):
        assert got['mano'][k].dtype == np.float32 and np.array_equal(got['mano'][k], assets['mano'][k]), k
    assert np.array_equal(got['anchor']['face_vert_idx'], assets['anchor']['face_vert_idx'])
    assert np.array_equal(got['anchor']['vert2joint'], assets['anchor']['vert2joint'])
    # anchor_weight.txt is text (make_golden writes %.9e: 9 significant digits round-trip a float32)
    assert np.array_equal(got['anchor']['anchor_weight'], assets['anchor']['anchor_weight'])
    assert list(got['ycb'].keys()) == list(assets['ycb'].keys())
    for n, v in assets['ycb'].items():
        for k in ('kpt3d', 'verts_sampled', 'CoM', 'verts', 'bbox3d'):
            assert np.array_equal(got['ycb'][n][k], v[k]), (n, k)
        assert got['ycb'][n]['diameter'] == v['diameter']
    assert 'SYNTHETIC' not in capsys.readouterr().err


def test_missing_tables_fall_back_to_synthetic_and_say_so(tmp_path, capsys):
    from vpho_amd import assets as A
    A._reported.clear()
    got = A.load_assets(str(tmp_path / 'nothing_here'))
    assert got['synthetic'] is True and got['sources'] == {'ycb': 'synthetic', 'anchor': 'synthetic', 'mano': 'synthetic'}
    err = capsys.readouterr().err
    assert err.count('SYNTHETIC') == 3 and 'MANO_RIGHT.pkl' in err and 'object_mesh_info.pkl' in err and 'anchor_weight.txt' in err
    A.load_assets(str(tmp_path / 'nothing_here'))
    assert capsys.readouterr().err == ''                      # once per process


@pytest.mark.parametrize('damage', ['truncated_mano', 'wrong_shape_anchor', 'missing_one_anchor_file', 'ycb_not_a_pickle', 'ycb_class_missing'])
def test_damaged_assets_raise_instead_of_turning_synthetic(tmp_path, assets, damage):
    from vpho_amd.assets import load_assets, AssetError
    root = _write_reference_layout(tmp_path, assets)
    if damage == 'truncated_mano':
        p = os.path.join(root, 'mano_v1_2', 'models', 'MANO_RIGHT.pkl')
        open(p, 'wb').write(open(p, 'rb').read()[:1000])
        match = 'mano'
    elif damage == 'wrong_shape_anchor':
        np.savetxt(os.path.join(root, '2021_CVPR_CPF', 'anchor', 'face_vertex_idx.txt'), np.zeros((31, 3), int), fmt='%d')
        match = r'face_vertex_idx.txt has shape \(31, 3\)'
    elif damage == 'missing_one_anchor_file':
        os.remove(os.path.join(root, 'ours', 'vert2joint.pkl'))
        match = 'incomplete asset set'
    elif damage == 'ycb_not_a_pickle':
        open(os.path.join(root, 'ours', 'object_mesh_info.pkl'), 'w').write('not a pickle')
        match = 'object_mesh_info.pkl'
    else:
        p = os.path.join(root, 'ours', 'object_mesh_info.pkl')
        mesh = pickle.load(open(p, 'rb'))
        del mesh[next(iter(mesh))]
        pickle.dump(mesh, open(p, 'wb'))
        match = 'lacks the classes'
    with pytest.raises(AssetError, match=match):
        load_assets(root)
