// raster_experiments.h -- CEILING builds of the edge sweeps: what would the iteration gain if a part of the kernel cost nothing?
// Every variant below produces WRONG gradients; they exist to bound an optimisation before it is built (EXPERIMENTS.md, rounds
// 4-5: tools/ab_state.py runs them at fixed states of a fit with lr = 0, so they see the shipped build's workload).
//   tools/ab_build.sh <name> -DHM_EXPERIMENT -DSWEEP_EXP=<n>
//     1  no stage 2 (items are classified, nothing is collected)          4  no pair rounds at all
//     2  metadata loads only (faces staged, nothing else)                 5  every lane flushes onto an accumulator of its own
//     3  at most one pair round per stage-2 trip                             (no same-address LDS atomics)
//                                                                         6  no LDS atomics in the pair loop
// The guards `eps > 0` / `eps < 0` keep the compiler from proving the skipped code dead at compile time in other kernels' paths.
#pragma once
#ifndef HM_EXPERIMENT
#error "raster_experiments.h holds timing-only variants with wrong results: build with -DHM_EXPERIMENT (never the release library)"
#endif
#if !defined(SWEEP_EXP) || SWEEP_EXP < 1 || SWEEP_EXP > 6
#error "-DHM_EXPERIMENT needs -DSWEEP_EXP=1..6"
#endif
#if SWEEP_EXP == 2
#define SWEEP_HOOK_PASS_STAGED(eps, nfp) if ((eps) > 0.f) { if ((nfp) < SWEEP_PASS_FACES) break; else continue; }
#endif
#if SWEEP_EXP == 1
#define SWEEP_HOOK_STAGE1_DONE(eps, qn) if ((eps) > 0.f) (qn) = 0
#endif
#if SWEEP_EXP == 3
#define SWEEP_HOOK_PAIR_ROUND(eps, base) if ((base) > 0 && (eps) > 0.f) break
#endif
#if SWEEP_EXP == 4
#define SWEEP_HOOK_PAIR_ROUND(eps, base) if ((eps) > 0.f) break
#endif
#if SWEEP_EXP == 5
#define SWEEP_HOOK_FLUSH(eps, fg, lane, cur, acc0, acc1) do { double* f_ = (fg)[(lane) & 15]; unsafeAtomicAdd(f_ + ((lane) >> 4), acc0); \
                                                              unsafeAtomicAdd(f_ + 4 + (((lane) >> 4) & 1), acc1); } while (0)
#endif
#if SWEEP_EXP == 6
#define SWEEP_HOOK_FLUSH(eps, fg, lane, cur, acc0, acc1) do { if ((eps) < 0.f) (fg)[0][0] = (acc0) + (acc1); } while (0)
#endif
