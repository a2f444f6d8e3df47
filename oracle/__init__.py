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
  * two forms of the same restatement, both pinned to the reference's goldens (``tests/test_oracle_golden.py``):
      - the FAITHFUL form: the reference's own torch expressions, autograd, ``torch.optim.Adam`` (``oracle.model`` with
        ``REFERENCE_FORM``, ``oracle.lbs``, ``oracle.jointopt.optimize_hand_object``);
      - the WRITTEN-OUT form: the same mathematics as ONE stated sequence of IEEE fp32 operations, so that an optimisation's
        end state is a defined quantity - the same for any number of host threads - and can be compared with the HIP loop
        bit for bit: ``oracle.objchain`` (object: order-independent sums), ``oracle.handchain`` (hand: MANO layer forward /
        backward, 2-D / smoothness / interaction / PCA terms, contact, collision), ``oracle.depthchain`` (ordinal depth
        term), ``oracle.posechain`` (pose initialisation), ``oracle.adam``; C in ``oracle/csrc/objchain.c`` and
        ``oracle/csrc/lbs_exact.c``; ``oracle.jointopt.reproducible_step`` / ``reproducible_step_shared_scale``.
        ``tests/test_objchain.py`` holds it within fp32 rounding of the faithful form.  One stated deviation: the nearest-
        vertex search differences coordinates before squaring (the reference expands |a|^2 + |b|^2 - 2ab, whose rounding
        names another neighbour in near-ties; ``test_written_out_step2_terms_equal_autograd``).
"""
