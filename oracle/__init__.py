"""oracle/ -- CPU restatement of the reference's joint-optimisation hot path.

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package, and only as the checker.  ``homan_amd`` never imports it.

Parity status (see DESIGN.md "Oracle"):
  * composition layer (homan/homan.py, losses.py, lossutils.py,
    interactions/*.py, utils/*.py, jointopt.py): restated in ``oracle.model``
    / ``oracle.jointopt`` and PINNED against the reference's own Python,
    imported in the build container over these same leaves
    (``tools/refharness``; vectors in ``tests/golden``).
  * leaves (`neural_renderer`, `sdf`, `mano`, `libyana`): third-party,
    un-vendored, un-pinned packages whose sources are not in /root/reference
    and for which the reference holds no tests or golden vectors ->
    PARITY UNPINNED at the leaves; ``oracle.nmr``, ``oracle.sdfgrid``,
    ``oracle.lbs`` restate the published algorithms and state every
    convention they fix.
"""
