"""Host logic of the training step that needs no GPU: the index maps of the input-gradient weight cache and the packed-layout hand-over
of convolution gradients into the flat gradient buffer."""
import torch


def _builders(cout, cin, kh, kw, pad):
    from vpho_amd import conv_backward as CB
    cp = cout + (4 - cout % 4) % 4
    out = {('s1', kh, kw): lambda w: CB._flip_transpose(CB._pad_rows4(w), cp, cin, kh, kw)}
    for py in (0, 1):
        for px in (0, 1):
            ty, tx = CB._phase_taps(py, kh, pad), CB._phase_taps(px, kw, pad)
            if ty and tx:
                out[('s2', kh, kw, pad, py, px)] = lambda w, ty=ty, tx=tx: CB._phase_weights(CB._pad_rows4(w), cin, kh, kw, ty, tx)
    return out


def test_dgrad_weight_cache_rebuilds_every_layout_with_one_gather():
    """Every derived layout (1x1 transpose, 3x3 flip + transpose, the stride-2 phases of 3x3 / 1x1 / 2x2 kernels, Cout padded to a
    multiple of 4) after the weights changed in place == the layout recipe applied to the new values; zero padding stays zero."""
    from vpho_amd import conv_backward as CB
    g = torch.Generator().manual_seed(3)
    cache = CB.DgradWeightCache()
    cases = [(8, 12, 1, 1, 0), (8, 4, 3, 3, 1), (21, 8, 1, 1, 0), (6, 4, 3, 3, 1), (12, 4, 2, 2, 0), (7, 5, 7, 7, 3)]
    weights, want = [], []
    for cout, cin, kh, kw, pad in cases:
        w = torch.randn(cout, kh * kw * cin, generator=g)
        weights.append(w)
        for key, b in _builders(cout, cin, kh, kw, pad).items():
            first = cache.get(w, key, b)
            assert torch.equal(first, b(w))
            want.append((w, key, b))
    for w in weights:                                       # an optimiser step: new values in place
        w.mul_(0.5).add_(torch.randn(w.shape, generator=g))
    cache.refresh()
    assert cache.flat.numel() == sum(b(w).numel() for w, _, b in want)
    for w, key, b in want:
        got = cache.get(w, key, b)
        assert got.data_ptr() >= cache.flat.data_ptr() and torch.equal(got, b(w)), key
    # a second refresh re-uses the maps (nothing new was registered)
    m = cache.map
    weights[0].zero_()
    cache.refresh()
    assert cache.map is m and float(cache.get(*want[0]).abs().max()) == 0.0


def test_dgrad_weight_cache_notices_an_unannounced_in_place_write():
    """a weight written in place WITHOUT refresh() (another optimiser, a finite-difference probe) must not leave the input
    gradients on the old layouts: get() compares the tensor's version counter with the one its layouts were built from and rebuilds."""
    from vpho_amd import conv_backward as CB
    g = torch.Generator().manual_seed(5)
    cache = CB.DgradWeightCache()
    w1, w2 = torch.randn(8, 9 * 4, generator=g), torch.randn(8, 12, generator=g)
    b1 = _builders(8, 4, 3, 3, 1)
    b2 = _builders(8, 12, 1, 1, 0)
    for w, bs in ((w1, b1), (w2, b2)):
        for key, b in bs.items():
            cache.get(w, key, b)
    cache.refresh()
    assert cache.rebuilds == 0
    w1.add_(1.0)                                            # nobody calls refresh()
    key, b = next(iter(b1.items()))
    assert torch.equal(cache.get(w1, key, b), b(w1)) and cache.rebuilds == 1
    for key, b in b1.items():                               # the one rebuild served every layout of every weight
        assert torch.equal(cache.get(w1, key, b), b(w1))
    assert cache.rebuilds == 1
    w2.mul_(2.0)
    key, b = next(iter(b2.items()))
    assert torch.equal(cache.get(w2, key, b), b(w2)) and cache.rebuilds == 2
    # a layout registered for the FIRST time after an unannounced write of an already known weight
    w1.sub_(0.25)
    extra = lambda w: CB._pad_rows4(w).clone()
    assert torch.equal(cache.get(w1, ('extra',), extra), extra(w1))
    key, b = next(iter(b1.items()))
    assert torch.equal(cache.get(w1, key, b), b(w1))


def test_cache_is_only_consulted_inside_its_context():
    from vpho_amd import conv_backward as CB
    w = torch.randn(4, 8)
    b = lambda t: t.t().contiguous()
    cache = CB.DgradWeightCache()
    assert CB._derived(w, ('s1', 1, 1), b).data_ptr() != CB._derived(w, ('s1', 1, 1), b).data_ptr() and not cache.entries
    with cache:
        a1 = CB._derived(w, ('s1', 1, 1), b)
        assert CB._derived(w, ('s1', 1, 1), b) is a1 and len(cache.entries) == 1
    assert CB._ACTIVE is None


def test_grad_buckets_take_packed_convolution_gradients():
    """A convolution gradient arrives as the reference-layout VIEW of its packed tensor (train_blocks._unpack_grad): the slot receives the
    packed tensor (contiguous -> multi-tensor copy); a plain reference-layout gradient from elsewhere is packed by one strided copy, padded
    input channels left zero."""
    from vpho_amd.grad_buckets import GradBuckets
    from vpho_amd.train_blocks import _unpack_grad
    cout, cin, kh, kw, cin_pad = 6, 3, 3, 3, 4
    meta = {'feature_extractor.layer0_h.0.weight': (cout, cin, kh, kw)}
    shapes = {'feature_extractor.layer0_h.0.weight': (cout, kh * kw * cin_pad), 'feature_extractor.layer0_h.1.bias': (cout,)}
    gb = GradBuckets(shapes, 'cpu', conv_meta=meta)
    gp = torch.randn(cout, kh * kw * cin_pad)
    gp.view(cout, kh, kw, cin_pad)[..., cin:] = 0
    view = _unpack_grad(gp, cout, cin, kh, kw)
    assert view.shape == (cout, cin, kh, kw) and view.packed_grad is gp
    bias = torch.randn(cout)
    gb.begin()
    gb.put({'feature_extractor.layer0_h.0.weight': view, 'feature_extractor.layer0_h.1.bias': bias})
    assert torch.equal(gb.view['feature_extractor.layer0_h.0.weight'], gp) and torch.equal(gb.view['feature_extractor.layer0_h.1.bias'], bias)
    gb.begin()
    gb.put({'feature_extractor.layer0_h.0.weight': view.clone()})                      # no packed tensor attached
    assert torch.equal(gb.view['feature_extractor.layer0_h.0.weight'], gp)
