"""Teacher-forced lock-step parity along the fused loop's trajectory at BASELINE cfg2 / cfg3 size (30 frames, 256x256,
3000-face bottle): BEFORE every step of the HIP loop its parameters are loaded into the CPU oracle, which evaluates that
step there (reference loop: homan/jointopt.py:158-192; bench.lockstep_parity).

Every step is a single-step comparison at identical parameters, so the chaos of the hard rasteriser - which makes FREE
trajectories separate after a few steps in the reference algorithm itself - cannot enter: what is
bounded here is the per-step error of the implementation, along the whole trajectory, at full size.

Bars: north_star's 1e-4 relative on every loss and 1e-3 mm on vertices, tightened to what the design guarantees - the
rotation, the rigid transform and the projection are evaluated in the oracle's operation order, so the object's vertices
and the face-index map of the raster are BIT-EQUAL (zero flipped samples, `loss_sil_obj` equal to the last bit up to the
summation order of the reduction) and the parameter gradients agree to 2e-5 of their largest entry."""
import os
import sys

import pytest

# `-m gpu` runs the 24- / 12-step legs (the driver's suite has a time limit, and the faithful oracle evaluates every step at full
# size at ~0.5 it/s); the 50- / 24-step legs of rounds 4-5 carry the marker `gpu_slow` (python -m pytest tests -m gpu_slow): same
# code, same bars.  The 400-step free runs are tools/chain_parity.py -> profiles/r0N_freerun_*.json.
def _lengths(short, full):
    return pytest.mark.parametrize("steps", [pytest.param(short, marks=pytest.mark.gpu), pytest.param(full, marks=pytest.mark.gpu_slow)])


ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _check(out, loss_bar, grad_bar=2e-5, loose=()):
    assert out["flipped_samples"] == 0, out["per_step"]
    assert out["object_vertices_bit_equal"]
    assert out["max_vert_diff_mm"]["object"] == 0.0
    # north_star: 1e-3 mm.  The MANO layer is ONE written-out operation order on both sides (oracle/csrc/lbs_exact.c <->
    # csrc/mano.hip, shared sin / cos): the hand's vertices are bit-equal too
    assert out["hand_vertices_bit_equal"] and out["max_vert_diff_mm"]["hand"] == 0.0, out["max_vert_diff_mm"]
    for k, v in out["worst_loss_per_key"].items():                                    # north_star: 1e-4
        assert v < (loose[k] if k in loose else loss_bar), (k, v)
    assert out["max_grad_err"] < grad_bar, out["worst_grad_per_step"]
    assert out["max_handobj_maxdist_abs_m"] < 1e-5                                    # see tests/test_model_gpu.py METRIC_ATOL
    for k, v in out["max_rel_metric"].items():
        if k != "handobj_maxdist":
            assert v < 1e-5, (k, v)


@_lengths(24, 50)
def test_lockstep_cfg2_full_size(steps, mano_model):
    sys.path.insert(0, ROOT)
    import bench_parity as bench
    out = bench.lockstep_parity(mano_model, step2=False, steps=steps, free_run=False)
    _check(out, 1e-5)
    assert out["worst_loss_per_key"]["loss_sil_obj"] < 1e-6


@_lengths(12, 24)
def test_lockstep_cfg2_with_the_depth_term_full_size(steps, mano_model):
    """cfg2 as BASELINE.json words it - sil / kp / DEPTH / smooth - at 30 frames x 256^2: the ordinal depth term of reference
    homan.py:384-419 / lossutils.py:133-169 (oracle-pinned: the reference's own call site raises) in the fused loop, every one of
    24 steps re-evaluated by the CPU oracle at the HIP parameters.  Losses incl. loss_depth within 1e-4, zero flipped samples in
    the silhouette raster and in both depth renders (full-image camera; object and hand vertices are bit-equal)."""
    sys.path.insert(0, ROOT)
    import bench_parity as bench
    out = bench.lockstep_parity(mano_model, step2=False, steps=steps, free_run=False, ordinal_depth=True)
    assert out["first_step_over_tol"] is None, out["per_step"]
    assert out["worst_loss_per_key"]["loss_depth"] < 1e-4, out["worst_loss_per_key"]
    assert out["flipped_samples"] == 0 and out["flipped_depth_samples"]["object"] == 0, (out["flipped_samples"], out["flipped_depth_samples"])
    assert out["flipped_depth_samples"]["hand"] == 0, out["flipped_depth_samples"]       # (bit-equal hand vertices)
    assert out["hand_vertices_bit_equal"]
    assert out["object_vertices_bit_equal"]
    assert out["max_grad_err"] < 2e-4, out["worst_grad_per_step"]


@_lengths(12, 50)
def test_lockstep_cfg3_full_size(steps, mano_model):
    sys.path.insert(0, ROOT)
    import bench_parity as bench
    out = bench.lockstep_parity(mano_model, step2=True, steps=steps, free_run=False)      # (the faithful-form leg is the independent check; the CPU side runs ~0.5 it/s on this set)
    # Losses: every term within 1e-5 of the faithful oracle's (measured 2.4e-7; `loss_collision` - a handful of trilinear SDF
    # samples, conditioned at ~1e-4 per ulp of a hand vertex - came down from 2e-4 to 1.4e-7 when the hand's vertices became
    # bit-equal).  Gradients: 5e-4 of the largest entry (measured 3.1e-4), all of it the contact term's NEAREST-VERTEX picks:
    # the faithful oracle ranks neighbours by the reference's |a|^2 + |b|^2 - 2ab (contactloss.py:60-79, rounding error ~4e-8 m^2
    # at |a|^2 ~ 0.36 m^2), the kernels by differenced coordinates, so a few hand vertices with two object vertices at almost
    # the same distance pull on the other one (tests/test_objchain.py::test_written_out_step2_terms_equal_autograd).  Against
    # the oracle's WRITTEN-OUT chain, which searches like the kernels, every gradient is bit-equal (tests/test_handchain_gpu.py).
    _check(out, 1e-5, grad_bar=5e-4)
    assert out["max_collision_rel_given_hip_vertices"] < 1e-5


@pytest.mark.gpu
def test_free_running_divergence_is_chaos_not_semantics(mano_model):
    """The FREE trajectories (HIP loop vs oracle loop with torch Adam) from identical inputs: the first step agrees to
    rounding, the first samples that differ appear only AFTER an optimiser step (never at step 0), and at the step before
    the 1e-4 band is left the lock-step comparison is still at rounding level - i.e. the separation is the algorithm's
    sensitivity to last-bit parameter differences (Adam normalises the step size), not a difference in what is computed."""
    sys.path.insert(0, ROOT)
    import bench_parity as bench
    out = bench.lockstep_parity(mano_model, step2=False, steps=14, frames=10, size=128, obj="cube", free_run=True)
    fr = out["free_run"]
    assert fr["samples_differing_per_step"][0] == 0
    assert fr["rel_diff_worst_per_step"][0] < 1e-6
    assert out["flipped_samples"] == 0 and out["max_grad_err"] < 2e-5
    if fr["first_step_over_tol"] is not None:
        assert fr["first_step_over_tol"] >= 2
        assert fr["before_separation"]["lockstep"] < 2e-5
