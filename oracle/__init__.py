"""CPU oracle for the VPHO per-image inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``vpho_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker / the timed CPU baseline.

Every function restates (in our own code, torch-CPU / numpy) the algorithm of
the reference file:line it cites (paths relative to the reference root).  The
third-party leaves whose source is not in the reference tree are restated from
their published algorithms:

* pytorch3d ``transforms/rotation_conversions.py`` (0.7.x, un-pinned upstream)
* manopth ``manolayer.py`` / ``rodrigues_layer.py`` / ``tensutils.py`` (un-pinned)
* torchvision 0.17.0 ``ops.roi_align`` (legacy ``aligned=False``)
* scipy 1.12 ``integrate.solve_ivp(method='RK45')``

Pinning: ``tests/golden/*.npz`` were produced by ``tests/golden/make_golden.py``
importing the reference's own Python modules (with the four leaves above bound
to these restatements, because those packages are not installed) and by the
installed scipy ``solve_ivp``; ``tests/test_oracle_golden.py`` checks the oracle
against them.  The three un-installed leaves themselves have no reference test
or fixture -> *parity unpinned* for roi_align / pytorch3d conversions / MANO LBS
beyond the closed-form known-answer tests in ``tests/test_oracle_leaves.py``.
"""
