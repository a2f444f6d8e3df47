// raster_hooks.h -- named points of the edge-sweep kernel where a MEASUREMENT build can cut work out.  The release build
// defines every hook empty; -DHM_EXPERIMENT pulls the definitions from raster_experiments.h (ceiling builds of tools/ab_build.sh:
// timing only, their results are wrong by construction and no test passes on them).
#pragma once
#ifdef HM_EXPERIMENT
#include "raster_experiments.h"
#endif
#ifndef SWEEP_HOOK_PASS_STAGED
#define SWEEP_HOOK_PASS_STAGED(eps, nfp)        // a pass's faces and family constants are in LDS
#endif
#ifndef SWEEP_HOOK_STAGE1_DONE
#define SWEEP_HOOK_STAGE1_DONE(eps, qn)         // stage 1 has queued the unit's reachable items
#endif
#ifndef SWEEP_HOOK_PAIR_ROUND
#define SWEEP_HOOK_PAIR_ROUND(eps, base)        // top of a round of 256 (item, source) pairs
#endif
// SWEEP_HOOK_FLUSH(eps, fg, lane, cur, acc0, acc1): replaces the flush of a lane's running sums into the face's LDS accumulators
