"""Static tables the hot path needs: MANO arrays, YCB object tables, CPF anchors, vert2joint.

The reference loads these from files that are not redistributable / not in its tree
(``asset/mano_v1_2/models/MANO_RIGHT.pkl`` head_mano.py:48-55; ``asset/ours/object_mesh_info.pkl``
dataset/base.py:204-258; ``asset/2021_CVPR_CPF/anchor/*`` physics_fn.py:185-199;
``asset/ours/vert2joint.pkl`` hand_fn.py:427-432).  ``load_assets`` reads the real files when they
exist under ``asset_root`` and otherwise falls back to ``synthetic_assets`` -- seeded arrays of identical
shape and plausible geometry so that synthetic benchmarks and parity tests exercise the
same code paths.
"""
import os
import pickle
import numpy as np

YCB_NAMES = [
    '002_master_chef_can', '003_cracker_box', '004_sugar_box', '005_tomato_soup_can', '006_mustard_bottle',
    '007_tuna_fish_can', '008_pudding_box', '009_gelatin_box', '010_potted_meat_can', '011_banana',
    '019_pitcher_base', '021_bleach_cleanser', '024_bowl', '025_mug', '035_power_drill', '036_wood_block',
    '037_scissors', '040_large_marker', '051_large_clamp', '052_extra_large_clamp', '061_foam_brick',
]

# MANO kinematic tree (parents) in MANO's own joint order: index, middle, pinky, ring, thumb.
MANO_PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]

# physics_fn.py:152-165 resolved: for anchor a, (joint_from, joint_to) in the 21-joint manopth order.
_SK = {0: [[0, 1], [0, 5], [0, 9], [0, 13], [0, 17]], 1: [[1, 2], [5, 6], [9, 10], [13, 14], [17, 18]],
       2: [[2, 3], [6, 7], [10, 11], [14, 15], [18, 19]], 3: [[3, 4], [7, 8], [11, 12], [15, 16], [19, 20]]}
_LABEL = [5, 12, 19, 18, 26, 25, 6, 0, 7, 13, 20, 27, 1, 8, 14, 21, 28,
          2, 3, 4, 9, 11, 10, 15, 17, 16, 22, 24, 23, 29, 31, 30]
_CORR = [_SK[0][1], _SK[0][2], _SK[0][3], _SK[0][3], _SK[0][4], _SK[0][4],
         _SK[0][0], _SK[0][0], _SK[1][1], _SK[1][2], _SK[1][3], _SK[1][4],
         _SK[2][0], _SK[2][1], _SK[2][2], _SK[2][3], _SK[2][4]] + \
        [_SK[3][f] for f in range(5) for _ in range(3)]
ANCHOR_SKELETON = np.array(_CORR, dtype=np.int64)[np.argsort(np.array(_LABEL))]  # (32,2)


def _unit(v):
    return v / (np.linalg.norm(v, axis=-1, keepdims=True) + 1e-12)


def synthetic_mano(rng):
    """Hand-like MANO tables: 16-joint skeleton, 778 vertices scattered around the bones."""
    # finger base positions / directions in MANO order (index, middle, pinky, ring, thumb), metres
    base = np.array([[0.095, 0.005, 0.025], [0.095, 0.003, 0.003], [0.080, -0.002, -0.038],
                     [0.090, 0.000, -0.018], [0.025, -0.010, 0.035]])
    direc = _unit(np.array([[1.0, 0.0, 0.08], [1.0, 0.0, 0.0], [1.0, 0.0, -0.15], [1.0, 0.0, -0.07], [0.6, -0.1, 0.7]]))
    seg = np.array([[0.032, 0.022, 0.020], [0.035, 0.025, 0.021], [0.025, 0.017, 0.017],
                    [0.032, 0.023, 0.020], [0.032, 0.028, 0.024]])
    J = np.zeros((16, 3))
    for f in range(5):
        p = base[f].copy()
        for k in range(3):
            J[1 + 3 * f + k] = p
            p = p + direc[f] * seg[f, k]
    tips = np.stack([J[3 + 3 * f] + direc[f] * seg[f, 2] for f in range(5)])
    # vertices: 778 points, each bound to a bone (joint j -> child or tip)
    nv = 778
    owner = np.concatenate([np.zeros(238, np.int64), np.repeat(np.arange(1, 16), 36)])
    assert owner.shape[0] == nv
    v = np.zeros((nv, 3))
    for i in range(nv):
        j = owner[i]
        if j == 0:
            u, w = rng.uniform(0, 1), rng.uniform(-1, 1)
            v[i] = np.array([0.09 * u, rng.uniform(-0.012, 0.012), 0.045 * w * (0.6 + 0.4 * u)])
        else:
            f, k = (j - 1) // 3, (j - 1) % 3
            end = J[j + 1] if k < 2 else tips[f]
            s = rng.uniform(0, 1)
            ang = rng.uniform(0, 2 * np.pi)
            d = direc[f]
            n1 = _unit(np.cross(d, np.array([0.0, 1.0, 0.0])))
            n2 = np.cross(d, n1)
            rad = 0.009 - 0.001 * k
            v[i] = J[j] * (1 - s) + end * s + rad * (np.cos(ang) * n1 + np.sin(ang) * n2)
    # make the 5 manopth tip vertices sit at the finger tips (thumb, index, middle, ring, pinky = 745,317,444,556,673)
    for vid, f in zip([745, 317, 444, 556, 673], [4, 0, 1, 3, 2]):
        v[vid] = tips[f]
        owner[vid] = 3 + 3 * f
    # skinning weights: owner + parent blend
    W = np.zeros((nv, 16))
    for i in range(nv):
        j = owner[i]
        a = rng.uniform(0.6, 1.0)
        W[i, j] = a
        p = MANO_PARENTS[j] if j > 0 else int(rng.choice([1, 4, 7, 10, 13]))
        W[i, p] += 1 - a
    # joint regressor: sparse convex combination of vertices, solved so that J_reg @ v == J approximately
    Jr = np.zeros((16, nv))
    for j in range(16):
        idx = np.argsort(np.linalg.norm(v - J[j], axis=1))[:12]
        w = rng.uniform(0.5, 1.5, size=12)
        Jr[j, idx] = w / w.sum()
    shapedirs = rng.normal(0, 0.0015, size=(nv, 3, 10)) * (1 + np.linalg.norm(v, axis=1)[:, None, None] * 8)
    posedirs = rng.normal(0, 0.0008, size=(nv, 3, 135))
    return dict(v_template=v.astype(np.float32), shapedirs=shapedirs.astype(np.float32),
                posedirs=posedirs.astype(np.float32), J_regressor=Jr.astype(np.float32),
                weights=W.astype(np.float32))


def synthetic_ycb(rng):
    """21 objects: random box; 27 key-points = 3x3x3 grid (misc_fn.py:42-67), 2048 verts on the box surface,
    CoM near the centre."""
    out = {}
    for name in YCB_NAMES:
        half = rng.uniform(0.025, 0.09, size=3)
        lo, hi = -half, half
        kpt = np.array([[lo[d] + (w / 2) * (hi[d] - lo[d]) for d, w in enumerate((i, j, k))]
                        for i in range(3) for j in range(3) for k in range(3)])
        def surf(n):
            p = rng.uniform(-1, 1, size=(n, 3))
            ax = rng.integers(0, 3, size=n)
            sg = rng.choice([-1.0, 1.0], size=n)
            p[np.arange(n), ax] = sg
            return p * half
        out[name] = dict(kpt3d=kpt.astype(np.float32), verts_sampled=surf(2048).astype(np.float32),
                         CoM=(rng.normal(0, 0.004, size=3)).astype(np.float32), verts=surf(2048).astype(np.float32))
        # evaluation tables (lib/dataset/base.py:235-244): the 8 box corners and the model diameter; no random draws
        out[name]['bbox3d'] = np.array([[sx * half[0], sy * half[1], sz * half[2]] for sx in (-1, 1) for sy in (-1, 1)
                                        for sz in (-1, 1)], dtype=np.float32)
        out[name]['diameter'] = float(2.0 * np.linalg.norm(half))
    return out


def synthetic_anchor(rng, mano):
    """32 CPF anchors: a face (3 nearby vertices) + 2 barycentric weights each; vert2joint (21,778)."""
    v = mano['v_template'].astype(np.float64)
    face = np.zeros((32, 3), np.int64)
    seeds = rng.choice(778, size=32, replace=False)
    for a, s in enumerate(seeds):
        nb = np.argsort(np.linalg.norm(v - v[s], axis=1))[:8]
        face[a] = [s] + list(rng.choice(nb[1:], size=2, replace=False))
    aw = rng.uniform(0.1, 0.45, size=(32, 2))
    v2j = np.zeros((21, 778))
    # joints of the 21-joint manopth order from the template: crude convex regressor around MANO joints / tips
    J16 = mano['J_regressor'].astype(np.float64) @ v
    tips = v[[745, 317, 444, 556, 673]]
    J21 = np.concatenate([J16, tips], 0)[[0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]]
    for j in range(21):
        idx = np.argsort(np.linalg.norm(v - J21[j], axis=1))[:10]
        w = rng.uniform(0.5, 1.5, size=10)
        v2j[j, idx] = w / w.sum()
    return dict(face_vert_idx=face, anchor_weight=aw.astype(np.float32), vert2joint=v2j.astype(np.float32))


def synthetic_assets(seed=0):
    rng = np.random.default_rng(seed)
    mano = synthetic_mano(rng)
    ycb = synthetic_ycb(rng)
    anchor = synthetic_anchor(rng, mano)
    return dict(mano=mano, ycb=ycb, anchor=anchor, synthetic=True)


class AssetError(RuntimeError):
    """an asset file is present but cannot be used (unreadable, wrong format, wrong shapes): never replaced by synthetic data silently"""


_reported = set()


def _report(msg):
    """one line per distinct message and process, on stderr"""
    if msg not in _reported:
        _reported.add(msg)
        import sys
        print('[vpho_amd.assets] ' + msg, file=sys.stderr, flush=True)


def _read(table, paths, parse):
    """``paths``: the files of one table.  None of them present -> None (the caller keeps the synthetic table and says so); all
    present -> parse(); some present, or parse() failing -> AssetError naming the table, the files and the cause."""
    have = [os.path.exists(p) for p in paths]
    if not any(have):
        return None
    if not all(have):
        raise AssetError(f'{table}: incomplete asset set, missing {[p for p, h in zip(paths, have) if not h]} (found '
                         f'{[p for p, h in zip(paths, have) if h]})')
    try:
        return parse()
    except AssetError:
        raise
    except Exception as e:
        raise AssetError(f'{table}: {paths} exist but cannot be parsed ({type(e).__name__}: {e})') from e


def _shape(table, name, arr, shape):
    if tuple(arr.shape) != tuple(shape):
        raise AssetError(f'{table}: {name} has shape {tuple(arr.shape)}, expected {tuple(shape)}')
    return arr


def load_assets(asset_root='asset', seed=0):
    """The three static tables of the model in the reference's on-disk formats, relative to ``asset_root`` (the reference opens them
    relative to the CWD at import time, quirk Q9):

    * ``mano_v1_2/models/MANO_RIGHT.pkl`` (manopth's ManoLayer, head_mano.py:48-55; unpickling the licensed file needs ``chumpy``)
    * ``2021_CVPR_CPF/anchor/{face_vertex_idx.txt, anchor_weight.txt}`` + ``ours/vert2joint.pkl`` (physics_fn.py:224-257,
      hand_fn.py:427-450)
    * ``ours/object_mesh_info.pkl`` (dataset/base.py:204-258: the cache the reference writes itself)

    A table none of whose files exist is replaced by the seeded synthetic one of the same shapes and REPORTED (stderr, once; and
    ``assets['sources'][table] == 'synthetic'``).  A file that exists but cannot be read or has the wrong shapes raises AssetError:
    a mis-placed or damaged asset must not turn into plausible-looking wrong results."""
    a = synthetic_assets(seed)
    src = {}
    j = lambda *p: os.path.join(asset_root, *p)

    def ycb():
        with open(j('ours', 'object_mesh_info.pkl'), 'rb') as f:
            mesh = pickle.load(f)
        out = {}
        for k, v in mesh.items():
            out[k] = dict(kpt3d=_shape('ycb', f'{k}.kpt3d', np.asarray(v['kpt3d'], np.float32), (27, 3)),
                          verts_sampled=_shape('ycb', f'{k}.verts_sampled', np.asarray(v['verts_sampled'], np.float32), (2048, 3)),
                          CoM=np.asarray(v['CoM'], np.float32).reshape(3), verts=np.asarray(v['verts'], np.float32).reshape(-1, 3),
                          bbox3d=_shape('ycb', f'{k}.bbox3d', np.asarray(v['bbox3d'], np.float32), (8, 3)), diameter=float(v['diameter']))
        missing = [n for n in YCB_NAMES if n not in out]
        if missing:
            raise AssetError(f'ycb: object_mesh_info.pkl lacks the classes {missing}')
        return out

    def anchor():
        root = j('2021_CVPR_CPF', 'anchor')
        face = _shape('anchor', 'face_vertex_idx.txt', np.loadtxt(os.path.join(root, 'face_vertex_idx.txt'), dtype=np.int64), (32, 3))
        aw = np.loadtxt(os.path.join(root, 'anchor_weight.txt')).astype(np.float32)
        if aw.ndim != 2 or aw.shape[0] != 32 or aw.shape[1] < 2:
            raise AssetError(f'anchor: anchor_weight.txt has shape {aw.shape}, expected (32, 2)')
        aw = aw[:, :2]                                        # physics_fn.py:124,248: a column of ones is prepended, columns 1 and 2 are used
        with open(j('ours', 'vert2joint.pkl'), 'rb') as f:
            v2j = _shape('anchor', 'vert2joint.pkl', np.asarray(pickle.load(f)['vert2joint'], np.float32), (21, 778))
        if face.min() < 0 or face.max() >= 778:
            raise AssetError('anchor: face_vertex_idx.txt indexes outside the 778 MANO vertices')
        return dict(face_vert_idx=face, anchor_weight=np.ascontiguousarray(aw), vert2joint=v2j)

    def mano():
        try:
            with open(j('mano_v1_2', 'models', 'MANO_RIGHT.pkl'), 'rb') as f:
                m = pickle.load(f, encoding='latin1')
        except ModuleNotFoundError as e:                       # the original file holds chumpy arrays
            raise AssetError(f'mano: MANO_RIGHT.pkl needs the module {e.name!r} to unpickle (the licensed file stores chumpy arrays); '
                             f'install it or re-save the five arrays as plain numpy') from e
        jr = m['J_regressor']
        jr = jr.toarray() if hasattr(jr, 'toarray') else jr
        return dict(v_template=_shape('mano', 'v_template', np.asarray(m['v_template'], np.float32), (778, 3)),
                    shapedirs=_shape('mano', 'shapedirs', np.asarray(m['shapedirs'], np.float32), (778, 3, 10)),
                    posedirs=_shape('mano', 'posedirs', np.asarray(m['posedirs'], np.float32), (778, 3, 135)),
                    J_regressor=_shape('mano', 'J_regressor', np.asarray(jr, np.float32), (16, 778)),
                    weights=_shape('mano', 'weights', np.asarray(m['weights'], np.float32), (778, 16)))

    for table, paths, parse in (
            ('ycb', [j('ours', 'object_mesh_info.pkl')], ycb),
            ('anchor', [j('2021_CVPR_CPF', 'anchor', 'face_vertex_idx.txt'), j('2021_CVPR_CPF', 'anchor', 'anchor_weight.txt'),
                        j('ours', 'vert2joint.pkl')], anchor),
            ('mano', [j('mano_v1_2', 'models', 'MANO_RIGHT.pkl')], mano)):
        got = _read(table, paths, parse)
        if got is None:
            src[table] = 'synthetic'
            _report(f'{table}: no file under {os.path.abspath(asset_root)!r} ({", ".join(os.path.relpath(p, asset_root) for p in paths)}) -> '
                    f'seeded SYNTHETIC table (seed {seed}); results are not those of the real assets')
        else:
            a[table] = got
            src[table] = paths[0] if len(paths) == 1 else os.path.dirname(paths[0])
    a['sources'] = src
    a['synthetic'] = all(v == 'synthetic' for v in src.values())
    return a
