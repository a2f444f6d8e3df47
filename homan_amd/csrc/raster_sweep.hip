// raster_sweep.hip -- backward, pass 2b / 3: edge sweeps over the flattened work list, vertex gather
#include "raster_ws.h"
#include "raster_hooks.h"

// ---------------------------------------------------------------- backward: shared helpers
__device__ __forceinline__ float sample_grad(const float* __restrict__ gimg, int S, int is, int xi, int yi)
{
    return 0.25f * gimg[(long)((is - 1 - yi) >> 1) * S + (xi >> 1)];
}

// contribution of sample d1 (pseudo-distance of the crossing to the two end points of the edge).  k0 / k1: the item's
// constants (c * 2) / is, folded once per item (sweep_item_scale) instead of three operations per pair and end point.
// IEEE divisions: the term is a defined function of its operands, the same float on the GPU and in the oracle.
__device__ __forceinline__ void sweep_term(float diff, int d1, float d1_cross, float k0, float k1, bool use0, bool use1,
                                           float eps, double magic, double& acc0, double& acc1)
{
    // straight-line (selects, no branches): every listed source has diff > 0 and nearly every item uses both end points, so
    // the conditions are almost always true and a taken branch costs more than the arithmetic it would skip.  A masked term
    // is an exact 0.
    const float t = (float)d1 - d1_cross;
    const bool live = diff > 0.0f;
    float dist0 = k0 * t, dist1 = k1 * t;
    dist0 += (0.0f < dist0) ? eps : -eps;          // (dist == 0 goes to -eps, like the reference's `0 < dist` test)
    dist1 += (0.0f < dist1) ? eps : -eps;
    const float g0 = diff / dist0, g1 = diff / dist1;
    acc0 -= hm_quant((live && use0) ? g0 : 0.0f, magic);
    acc1 -= hm_quant((live && use1) ? g1 : 0.0f, magic);
}
__device__ __forceinline__ float sweep_item_scale(float c, float two_over_is, bool pow2, int is)
{
    // (c * 2) / is; for a power of two the product with the exact inverse is the same float
    return pow2 ? (c * 2.0f) * two_over_is : (c * 2.0f) / (float)is;
}

// One workgroup (256 threads): the forward raster's launch order for the next iteration = its entries sorted by the time their
// workgroups took in this one, longest first (counting sort on 40 ns units, 1024 bins; the order inside a bin is whatever the
// LDS atomics give: scheduling only, results do not depend on the order).  wo_dyn (n) is rewritten through wo_tmp (n).
__device__ __forceinline__ void raster_reorder(int* __restrict__ wo_dyn, int* __restrict__ wo_tmp,
                                               const unsigned int* __restrict__ wg_cost, unsigned int* __restrict__ dyn_flag, int n,
                                               unsigned* __restrict__ s_hist)       // 1024 words of the caller's LDS
{
    __shared__ unsigned s_wsum[4];
    __shared__ unsigned s_zero;          // entries of idle workgroups (time 0: most of a clip's regions are background): they go
                                         // last in any order, placed by wave ballots - thousands of LDS atomics on ONE bin serialise
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 1024; i += 256) s_hist[i] = 0u;
    if (tid == 0) s_zero = 0u;
    __syncthreads();
    // (16 loads in flight per thread and pass: a dependent load per entry would make this workgroup the launch's tail)
    for (int base = 0; base < n; base += 256 * 16) {
        unsigned cv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int i = base + 256 * k + tid; cv[k] = i < n ? wg_cost[i] : 0u; }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (base + 256 * k + tid < n && cv[k] != 0u) atomicAdd(&s_hist[1023 - min(cv[k] >> 2, 1023u)], 1u);
    }
    __syncthreads();
    // exclusive prefix over the bins: four consecutive bins per thread, wave scan, wave totals
    unsigned c[4], tot = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[k] = s_hist[4 * tid + k]; tot += c[k]; }
    const unsigned incl = (unsigned)hm_wave_scan_incl((int)tot);
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    unsigned base0 = incl - tot;
    for (int q = 0; q < wv; ++q) base0 += s_wsum[q];
    const unsigned n_busy = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) { s_hist[4 * tid + k] = base0; base0 += c[k]; }
    __syncthreads();
    for (int base = 0; base < n; base += 256 * 16) {
        unsigned cv[16];
        int ev[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = base + 256 * k + tid;
            cv[k] = i < n ? wg_cost[i] : 0u;
            ev[k] = i < n ? wo_dyn[i] : 0;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const bool in = base + 256 * k + tid < n, zero = in && cv[k] == 0u;
            const unsigned long long zb = __ballot(zero);
            unsigned zbase = 0u;
            if (lane == 0 && zb) zbase = atomicAdd(&s_zero, (unsigned)__popcll(zb));
            zbase = (unsigned)__builtin_amdgcn_readfirstlane((int)zbase);
            if (zero) wo_tmp[n_busy + zbase + (unsigned)__popcll(zb & ((1ull << lane) - 1ull))] = ev[k];
            else if (in) wo_tmp[atomicAdd(&s_hist[1023 - min(cv[k] >> 2, 1023u)], 1u)] = ev[k];
        }
    }
    __threadfence();
    __syncthreads();
    for (int base = 0; base < n; base += 256 * 16) {
        int ev[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int i = base + 256 * k + tid; ev[k] = i < n ? __builtin_nontemporal_load(wo_tmp + i) : 0; }
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int i = base + 256 * k + tid; if (i < n) wo_dyn[i] = ev[k]; }
    }
    if (tid == 0) *dyn_flag = 1u;
}

// ---------------------------------------------------------------- backward, pass 2b: edge sweeps (see the work list above)
#ifdef SWEEP_STATS
__device__ unsigned long long g_sweep_n[16];     // stage 2: items, geo, act0, act1, on0, on1, pairs, trips; stage 1: items, geo, reach; pair rounds
#endif
#ifdef SWEEP_UNIT_PROFILE
__device__ int g_unit_prof[65536][4];            // per unit: wall-clock ticks, face passes, stage-2 trips | queued items << 8, pair rounds
#endif
// geometry of item j of face record fc: which (winding, edge, axis) family, which line d0r, where the edge crosses it
struct SweepGeo {
    int var, edge, axis, d0r, dir, a_in, a_out;
    float p00, p01, p10, p11, p20, p21, num, d1_cross;
    bool geo;
};
// Per (face of the pass, family) constants, built once per pass by the wave (k_bwd_sweep): the slope of the edge along the
// line axis (the IEEE division every item of the family used to repeat), the family's first line and the order of its two
// end points.  {slope bits, d0_from | (p00 < p10) << 16}
struct SweepLite { int var, axis, d0r, a_in, a_out; bool geo, pos; };
__device__ __forceinline__ SweepLite sweep_item_lite(const SweepFace& fc, const int2* __restrict__ famtab, int j, bool mine, int is)
{
    SweepLite q;
    int fam = 0;
#pragma unroll
    for (int stp = 8; stp > 0; stp >>= 1)
        if ((int)fc.cum[fam + stp - 1] <= j) fam += stp;
    const int fstart = fam ? (int)fc.cum[fam - 1] : 0;
    q.var = fam >= 6 ? 1 : 0;
    const int ci = fam - 6 * q.var;
    q.axis = ci & 1;
    const int edge = ci >> 1, v0 = q.var ? 2 - edge : edge;
    const float p00 = (q.axis ? fc.py : fc.px)[v0], p01 = (q.axis ? fc.px : fc.py)[v0];
    const int2 ft = famtab[fam];
    const float slope = __int_as_float(ft.x);
    const bool lt = (ft.y >> 16) & 1;
    const int dir = q.axis == 0 ? (lt ? -1 : 1) : (lt ? 1 : -1);
    q.d0r = mine ? (ft.y & 0xffff) + (j - fstart) : 0;
    const float d1_cross = slope * ((float)q.d0r - p00) + p01;
    bool geo = mine && d1_cross > -8.0f && d1_cross < (float)is + 8.0f;
    const int d1_in = geo ? ((dir > 0) ? (int)floorf(d1_cross) : (int)ceilf(d1_cross)) : 0;
    const int d1_out = d1_in + dir;
    geo = geo && !(d1_in < 0 || is <= d1_in || d1_out < 0 || is <= d1_out);
    q.a_in = geo ? d1_in : 0;
    q.a_out = geo ? d1_out : 0;
    q.geo = geo;
    q.pos = dir > 0;
    return q;
}
__device__ __forceinline__ SweepGeo sweep_item_geo(const SweepFace& fc, int j, bool mine, int is,
                                                   const int2* __restrict__ famtab = nullptr)
{
    SweepGeo q;
    int fam = 0;
#pragma unroll
    for (int stp = 8; stp > 0; stp >>= 1)
        if ((int)fc.cum[fam + stp - 1] <= j) fam += stp;      // fam + stp - 1 <= 11, and cum[11] > j
    const int fstart = fam ? (int)fc.cum[fam - 1] : 0;
    q.var = fam >= 6 ? 1 : 0;
    const int ci = fam - 6 * q.var;
    q.edge = ci >> 1;
    q.axis = ci & 1;
    int v0 = q.edge, v1 = q.edge == 2 ? 0 : q.edge + 1, v2 = q.edge == 0 ? 2 : q.edge - 1;
    if (q.var) { v0 = 2 - v0; v1 = 2 - v1; v2 = 2 - v2; }
    const float* pa = q.axis ? fc.py : fc.px;      // coordinate along which the lines are counted
    const float* pb = q.axis ? fc.px : fc.py;      // coordinate along the line
    q.p00 = pa[v0]; q.p01 = pb[v0]; q.p10 = pa[v1]; q.p11 = pb[v1]; q.p20 = pa[v2]; q.p21 = pb[v2];
    if (q.axis == 0) q.dir = (q.p00 < q.p10) ? -1 : 1;
    else q.dir = (q.p00 < q.p10) ? 1 : -1;
    const int d0_from = (int)fmaxf(ceilf(fminf(q.p00, q.p10)), 0.0f);
    q.num = q.p10 - q.p00;
    const float slope = famtab ? __int_as_float(famtab[fam].x) : (q.p11 - q.p01) / q.num;     // (the same float either way)
    q.d0r = mine ? d0_from + (j - fstart) : 0;
    q.d1_cross = slope * ((float)q.d0r - q.p00) + q.p01;
    bool geo = mine && q.d1_cross > -8.0f && q.d1_cross < (float)is + 8.0f;
    const int d1_in = geo ? ((q.dir > 0) ? (int)floorf(q.d1_cross) : (int)ceilf(q.d1_cross)) : 0;
    const int d1_out = d1_in + q.dir;
    geo = geo && !(d1_in < 0 || is <= d1_in || d1_out < 0 || is <= d1_out);
    q.a_in = geo ? d1_in : 0;
    q.a_out = geo ? d1_out : 0;
    q.geo = geo;
    return q;
}

// Accumulators that many lanes are about to add onto the SAME LDS word (a long sweep: dozens of consecutive lanes carry the same
// (face, corner) key) are combined per row of 16 lanes first: when at least SWEEP_COMBINE_MIN lanes of the wave flush at once,
// every row whose 16 lanes all flush the same key adds its accumulators with a DPP tree (exact sums: any order, see hm_quant) and
// leaves the flush to its lane 15 - 4 atomics per wave and address instead of 64 serialised ones.  Other rows flush lane by lane.
#ifndef SWEEP_COMBINE_MIN
#define SWEEP_COMBINE_MIN 32
#endif
#ifndef SWEEP_COMBINE_PAIRS
#define SWEEP_COMBINE_PAIRS 192     // the flush at the end of a trip looks only from that many pairs on (steady state of a fit: ~100 per trip)
#endif
__device__ __forceinline__ double sweep_row_sum(double v)          // lane 15 of every row: the row's total
{
    v += hm_dpp_f64<0xb1, 0xf>(v);
    v += hm_dpp_f64<0x4e, 0xf>(v);
    v += hm_dpp_f64<0x114, 0xf>(v);
    v += hm_dpp_f64<0x118, 0xf>(v);
    return v;
}
__device__ __forceinline__ void sweep_combine_rows(int& cur, double& acc0, double& acc1, bool flush, int lane)
{
    const unsigned long long nb = __ballot(flush);
    if (__popcll(nb) < SWEEP_COMBINE_MIN) return;                 // (wave-uniform)
    const int prev = __builtin_amdgcn_update_dpp(cur, cur, 0x111, 0xf, 0xf, false);      // row_shr:1 (lane 0 of a row: its own)
    const unsigned long long same = __ballot(prev == cur);
    const int sh = lane & 48;
    const bool uni = ((unsigned)(nb >> sh) & (unsigned)(same >> sh) & 0xffffu) == 0xffffu;
    const double r0 = sweep_row_sum(acc0), r1 = sweep_row_sum(acc1);
    if (uni) {
        if ((lane & 15) == 15) { acc0 = r0; acc1 = r1; }
        else { cur = -1; acc0 = 0.0; acc1 = 0.0; }
    }
}

// A wave takes UNITS of 256 consecutive items.  Per unit and per pass of <= 16 faces:
//   stage 1  every item (64 per trip): family + line geometry, then ONE 8-byte load of its line's summary {first / last set
//            position of both planes} - in the steady state of a fit ~70 % of the items have no source their sweeps could
//            reach (the bands of disagreement between render and target are thin) and stop here; the others are queued;
//   stage 2  the queued items, 64 per trip on full waves: owner tests at the edge, line records, the source slices of the
//            two sweeps, and the (item, source) pairs flattened over the wave as before.
// Filtering before the expensive half is what the 256-item unit is for: a 64-item unit leaves ~19 survivors, a quarter of a
// wave, and a divergent early-out saves nothing.  Unit composition depends only on the compaction block the items come from.
#ifndef SWEEP_WAVES_EU
#define SWEEP_WAVES_EU 5
#endif
template <bool W32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SWEEP_WAVES_EU, 8))) void k_bwd_sweep(SweepList sl, const int* __restrict__ idx_map,
                                                   const SweepSrc* __restrict__ srcs,
                                                   const uint4* __restrict__ lrec, int B, int F, int S,
                                                   float eps, double magic, double* __restrict__ parts,
                                                   const unsigned short* __restrict__ lsum,
                                                   const unsigned short* __restrict__ alpha16,
                                                   const unsigned int* __restrict__ ts_flag,
                                                   unsigned long long* __restrict__ ts_slots, int nsort,
                                                   int* __restrict__ wo_dyn, int* __restrict__ wo_tmp,
                                                   const unsigned int* __restrict__ wg_cost,
                                                   unsigned int* __restrict__ dyn_flag, int n_wo)
{
    // LDS copies are padded to an ODD number of dwords (17 / 9): lanes reading the same field of different faces / items
    // then fall into different banks (64- and 32-byte strides put every second / fourth element on the same bank)
    HM_CHAIN_KERNEL();
    typedef typename HmOff<W32>::t OFF;
    const unsigned long long ts_t0 = (unsigned long long)wall_clock64();
    const bool ts_on = hm_ts_enabled(ts_flag) && (threadIdx.x & 63) == 0;       // every wave: they walk their units independently
    if (ts_on) hm_ts_store(ts_slots, (long)blockIdx.x * 4 + (threadIdx.x >> 6), 0, ts_t0);
    struct FaceLds { SweepFace f; int pad; };
    struct ItemLds { SweepItem it; int pad; };
    __shared__ FaceLds s_face[4][SWEEP_PASS_FACES];
    __shared__ double s_fg[4][SWEEP_PASS_FACES][6];
    __shared__ int s_start[4][64];
    __shared__ int s_head[4][256];
    __shared__ ItemLds s_item[4][64];
    __shared__ unsigned short s_q[4][SWEEP_UNIT];
    __shared__ int2 s_fam[4][SWEEP_PASS_FACES][12];      // per (face of the pass, family): see sweep_item_lite
    __shared__ int s_fb[4][SWEEP_PASS_FACES][2];         // per (face, axis): range of the inward sweeps, lo | hi << 16
    // (optional workgroup 0: the forward raster's launch order for the next iteration, see raster_reorder - it rides this launch,
    //  the longest of the backward, so that it is nobody's tail; its histogram lives in s_head)
    if (nsort && blockIdx.x == 0) {
        raster_reorder(wo_dyn, wo_tmp, wg_cost, dyn_flag, n_wo, reinterpret_cast<unsigned*>(&s_head[0][0]));
        if (ts_on) hm_ts_store(ts_slots, (long)blockIdx.x * 4 + (threadIdx.x >> 6), 1, (unsigned long long)wall_clock64());
        return;
    }
    const int wid = (int)blockIdx.x - nsort, nwork = (int)gridDim.x - nsort;      // worker index / count (a multiple of 8)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int is = 2 * S;
    const bool pow2 = (is & (is - 1)) == 0;
    const float inv_is = 1.0f / (float)is;      // exact for powers of two
    const int wpl = is / 64;                    // 64-bit mask words per line
    const unsigned long long tot = sl.total[0];
    const int N = (int)(tot & 0xffffffffull), W = (int)(tot >> 32);
    const int U = (N + SWEEP_UNIT - 1) >> SWEEP_USHIFT;
    // XCD-aware unit assignment: workgroups are dealt to the 8 XCDs round-robin and every XCD has its own L2, while the
    // item list is frame-major.  Each XCD therefore takes one contiguous eighth of the units (~ B/8 whole frames): the
    // index-map lines, line records and source slices of a frame are then fetched into ONE L2 instead of eight (speed
    // only: nothing depends on where a workgroup really runs).  gridDim.x is a multiple of 8.
    const int xcd = wid & 7;
    const int xwaves = (nwork >> 3) * 4;
    const int u_end = (int)(((long)U * (xcd + 1)) >> 3);
    // (a unit's first face comes from the unit table; the wave requests the NEXT unit's entry together with the current
    //  unit's records, so only a wave's first unit pays that round trip)
    int first_ahead = -1;
    for (int u = __builtin_amdgcn_readfirstlane((int)(((long)U * xcd) >> 3) + (wid >> 3) * 4 + wv); u < u_end;
         u += xwaves) {
        const int ubeg = u << SWEEP_USHIFT, uend = ubeg + SWEEP_UNIT;
#ifdef SWEEP_UNIT_PROFILE
        const unsigned long long up_t0 = wall_clock64();
        int up_pass = 0, up_trips = 0, up_q = 0, up_rounds = 0;
#endif
        int first = first_ahead;
        if (first < 0) {
            if (u < sl.ucap) first = (int)sl.ufirst[u];
            else {                                   // beyond the unit table: last face with off <= first item of the unit
                int lo = 0, hi = W - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (sl.offs[mid] <= ubeg) lo = mid; else hi = mid - 1;
                }
                first = lo;
            }
            first = __builtin_amdgcn_readfirstlane(first);
        }
        const int u_ahead = u + xwaves;
        int ahead_v = -1;
        if (u_ahead < u_end && u_ahead < sl.ucap) ahead_v = (int)sl.ufirst[u_ahead];
        for (int fb = 0;; fb += SWEEP_PASS_FACES) {
            // first items of the pass's faces and of the face behind them (lane 16): sorted, so the faces inside the unit
            // are a prefix
            // (the records of the 16 faces that MAY belong to the pass are requested with their first items, not after them:
            //  one round trip; the rows of faces beyond the unit are dropped)
            const int o = (lane <= SWEEP_PASS_FACES && first + fb + lane < W) ? sl.offs[first + fb + lane] : 0x7fffffff;
            int row[SWEEP_PASS_FACES / 4];
#pragma unroll
            for (int k = 0; k < SWEEP_PASS_FACES / 4; ++k) {
                const int fi = first + fb + 4 * k + (lane >> 4);
                row[k] = fi < W ? reinterpret_cast<const int*>(sl.tab + fi)[lane & 15] : 0;
            }
            if (fb == 0) first_ahead = __builtin_amdgcn_readfirstlane(ahead_v);
            const int nfp = __popcll(__ballot(lane < SWEEP_PASS_FACES && o < uend));
            if (nfp == 0) break;
#ifdef SWEEP_UNIT_PROFILE
            ++up_pass;
#endif
            const int it_lo = max(ubeg, __builtin_amdgcn_readlane(o, 0));
            const int it_hi = min(min(uend, N), __builtin_amdgcn_readlane(o, nfp));     // (lane nfp: next face, or "none")
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < SWEEP_PASS_FACES / 4; ++k) {
                const int ent = 4 * k + (lane >> 4);
                if (ent < nfp) reinterpret_cast<int*>(&s_face[wv][ent].f)[lane & 15] = row[k];
            }
            for (int i = lane; i < SWEEP_PASS_FACES * 6; i += 64) (&s_fg[wv][0][0])[i] = 0.0;
            wave_sync();
            // family constants of the pass's faces: one division per (face, family) instead of one per item
            for (int idx = lane; idx < nfp * 12; idx += 64) {
                const int e = idx / 12, fam = idx - 12 * e;
                const SweepFace& f = s_face[wv][e].f;
                const int var = fam >= 6 ? 1 : 0, ci = fam - 6 * var, edge = ci >> 1, axis = ci & 1;
                int v0 = edge, v1 = edge == 2 ? 0 : edge + 1;
                if (var) { v0 = 2 - v0; v1 = 2 - v1; }
                const float* pa = axis ? f.py : f.px;
                const float* pb = axis ? f.px : f.py;
                const float p00 = pa[v0], p10 = pa[v1], p01 = pb[v0], p11 = pb[v1];
                const float slope = (p11 - p01) / (p10 - p00);
                const int d0_from = (int)fmaxf(ceilf(fminf(p00, p10)), 0.0f);
                s_fam[wv][e][fam] = make_int2(__float_as_int(slope), (d0_from & 0xffff) | ((p00 < p10) ? 1 << 16 : 0));
            }
            if (lane < 2 * nfp) {
                // the inward sweep stays inside the triangle: its extent along the line bounds the range
                const int e = lane >> 1, axis = lane & 1;
                const SweepFace& f = s_face[wv][e].f;
                const float* pb = axis ? f.px : f.py;
                const float tmin = fminf(pb[0], fminf(pb[1], pb[2])), tmax = fmaxf(pb[0], fmaxf(pb[1], pb[2]));
                s_fb[wv][e][axis] = max(0, (int)floorf(fmaxf(tmin, 0.f)) - 1) | (min(is - 1, (int)ceilf(fminf(tmax, (float)is)) + 1) << 16);
            }
            wave_sync();
            SWEEP_HOOK_PASS_STAGED(eps, nfp);
            // ---------------- stage 1: which items have a source in reach?  Four trips cover the unit; the summary loads of
            // all of them are in flight before the first is tested (one dependent round trip per unit, not per trip)
            int qn = 0;
            {
              for (int tb = 0; tb < SWEEP_TRIPS; tb += SWEEP_TBATCH) {      // (<= 4 trips' loads in flight at a time: registers)
                uint4 sm[SWEEP_TBATCH];
                // (per trip, packed - the four trips' state lives in registers until their loads have landed:
                //  s_io = a_in | pos << 12 | geo << 13, s_lohi = lo | hi << 16 of the inward range)
                int s_io[SWEEP_TBATCH], s_lohi[SWEEP_TBATCH], s_ent[SWEEP_TBATCH], s_own[SWEEP_TBATCH], s_fn[SWEEP_TBATCH];
                unsigned short s_aw[SWEEP_TBATCH];
#pragma unroll
                for (int t = 0; t < SWEEP_TBATCH; ++t) {
                    const int g = it_lo + 64 * (tb + t) + lane;
                    int el = -1;                                      // my face: last one of the pass with off <= g
                    for (int i = 0; i < nfp; ++i) el += (__builtin_amdgcn_readlane(o, i) <= g) ? 1 : 0;
                    const SweepFace& fc = s_face[wv][max(el, 0)].f;
                    // (past the last face of a compaction block: padding that belongs to no face)
                    const bool mine = g < it_hi && el >= 0 && g - fc.off < (int)fc.cum[11];
                    const SweepLite q = sweep_item_lite(fc, s_fam[wv][max(el, 0)], mine ? g - fc.off : 0, mine, is);
                    sm[t] = *reinterpret_cast<const uint4*>(hm_at<W32>(lsum, (((OFF)fc.b * 2 + q.axis) * is + q.d0r) * 8));
                    // the two samples at the edge, requested with the summary (one round trip): the owner of the sample just
                    // inside (the outward sweep runs only from a sample this winding owns) and the alpha word of the sample
                    // just outside (the inward sweep only from an empty one)
                    {
                        const int xi_in = q.axis ? q.a_in : q.d0r, yi_in = q.axis ? q.d0r : q.a_in;
                        const int xi_out = q.axis ? q.a_out : q.d0r, yo = is - 1 - (q.axis ? q.d0r : q.a_out);
                        s_own[t] = *hm_at<W32>(idx_map, ((OFF)fc.b * is + yi_in) * is + xi_in);
                        s_aw[t] = *hm_at<W32>(alpha16, ((((OFF)fc.b * (is >> 4)) + (yo >> 4)) * (is >> 4) + (xi_out >> 4)) * 16 + (yo & 15));
                        s_aw[t] = (unsigned short)((s_aw[t] >> (xi_out & 15)) & 1u);
                        s_fn[t] = fc.bf - fc.b * F + q.var * F;
                    }
                    s_io[t] = q.a_in | (q.pos ? 1 << 12 : 0) | (q.geo ? 1 << 13 : 0);
                    {
                        const int fbv = s_fb[wv][max(el, 0)][q.axis];
                        s_lohi[t] = q.pos ? ((fbv & 0xffff) | (q.a_in << 16)) : (q.a_in | (fbv & 0xffff0000));
                    }
                    s_ent[t] = (g - ubeg) | (max(el, 0) << SWEEP_USHIFT);
                }
#pragma unroll
                for (int t = 0; t < SWEEP_TBATCH; ++t) {
                    const int min0 = (int)(sm[t].x & 0xffffu), end0 = (int)(sm[t].x >> 16);
                    const int min1 = (int)(sm[t].z & 0xffffu), end1 = (int)(sm[t].z >> 16);
                    const bool pos = (s_io[t] >> 12) & 1, geo_t = (s_io[t] >> 13) & 1;
                    const int a_in_t = s_io[t] & 0xfff, a_out_t = a_in_t + (pos ? 1 : -1);
                    const int lo_t = s_lohi[t] & 0xffff, hi_t = s_lohi[t] >> 16;
                    // outward: plane 0 from the sample just outside the edge to the border (exact)
                    const bool out_ok = pos ? end0 > a_out_t : min0 <= a_out_t;
                    // inward: plane 1 inside [lo, hi] (first / last position, then the 64-sample words in between)
                    const int wlo = lo_t >> 6, whi = hi_t >> 6;
                    const bool in_ok = lo_t <= hi_t && min1 <= hi_t && end1 > lo_t &&
                                       ((sm[t].w >> wlo) & ((2u << (whi - wlo)) - 1u)) != 0u;
                    const bool a0 = geo_t && s_own[t] == s_fn[t] && out_ok;       // exact: the outward sweep has pairs
                    const bool a1 = geo_t && s_aw[t] == 0 && in_ok;                // (the inward range is refined in stage 2)
                    const bool reach = a0 || a1;
                    const unsigned long long bal = __ballot(reach);
#ifdef SWEEP_STATS
                    const unsigned long long gbal = __ballot(geo_t);
                    if (lane == 0) {
                        atomicAdd(&g_sweep_n[8], (unsigned long long)max(0, min(64, it_hi - (it_lo + 64 * (tb + t)))));
                        atomicAdd(&g_sweep_n[9], (unsigned long long)__popcll(gbal));
                        atomicAdd(&g_sweep_n[10], (unsigned long long)__popcll(bal));
                    }
                    {
                        const unsigned long long b12 = __ballot(geo_t && s_own[t] == s_fn[t]), b13 = __ballot(a0);
                        const unsigned long long b14 = __ballot(geo_t && s_aw[t] == 0), b15 = __ballot(a1);
                        if (lane == 0) {
                            atomicAdd(&g_sweep_n[12], (unsigned long long)__popcll(b12)); atomicAdd(&g_sweep_n[13], (unsigned long long)__popcll(b13));
                            atomicAdd(&g_sweep_n[14], (unsigned long long)__popcll(b14)); atomicAdd(&g_sweep_n[15], (unsigned long long)__popcll(b15));
                        }
                    }
#endif
                    if (reach) s_q[wv][qn + __popcll(bal & ((1ull << lane) - 1ull))] =
                                   (unsigned short)(s_ent[t] | (a0 ? 1 << 14 : 0) | (a1 ? 1 << 15 : 0));
                    qn += __popcll(bal);
                }
              }
            }
            wave_sync();
            SWEEP_HOOK_STAGE1_DONE(eps, qn);
            // ---------------- stage 2: the items that may collect something, 64 per trip
#ifdef SWEEP_UNIT_PROFILE
            up_q += qn;
#endif
            for (int s0 = 0; s0 < qn; s0 += 64) {
#ifdef SWEEP_UNIT_PROFILE
            ++up_trips;
#endif
            bool mine = s0 + lane < qn;
            const int ent = mine ? (int)s_q[wv][s0 + lane] : 0;
            const int g = ubeg + (ent & (SWEEP_UNIT - 1)), el = (ent >> SWEEP_USHIFT) & 15;
            const SweepFace& fc = s_face[wv][el].f;
            const SweepGeo q = sweep_item_geo(fc, mine ? g - fc.off : 0, mine, is, s_fam[wv][el]);
            const int var = q.var, edge = q.edge, axis = q.axis, d0r = q.d0r, dir = q.dir, a_in = q.a_in, a_out = q.a_out;
            const float p00 = q.p00, p01 = q.p01, p10 = q.p10, p11 = q.p11, p20 = q.p20, p21 = q.p21, num = q.num;
            const float d1_cross = q.d1_cross;
            const bool geo = q.geo;
            const int b = fc.b, fn = fc.bf - b * F + var * F;
            const bool use0 = p10 != (float)d0r, use1 = p00 != (float)d0r;
            // (IEEE divisions, like c2 below: every operand of a term is a defined function of the face and the line)
            const float c0 = use0 ? num / (p10 - (float)d0r) : 0.f;
            const float c1 = use1 ? num / ((float)d0r - p00) : 0.f;
            const bool act0 = geo && (ent & (1 << 14));   // outward: my own sample just inside the edge      (stage 1 looked
            const bool act1 = geo && (ent & (1 << 15));   // inward: only if the sample just outside is empty   both up)
            // [0] outward, from the sample just outside the edge to the border; [1] inward, across the triangle
            int rfrom[2], rto[2];
            {
                const int lim = (dir > 0) ? is - 1 : 0;
                rfrom[0] = max(min(a_out, lim), 0);
                rto[0] = min(max(a_out, lim), is - 1);
            }
            rfrom[1] = 0;
            rto[1] = -1;
            if (act1) {                                   // (silhouette edges only: a few per cent of the items)
                // crossing of the line with the other edge it meets: one division on selected operands
                const bool far02 = ((float)d0r - p00) * ((float)d0r - p20) < 0.0f;
                const float na = far02 ? p21 - p01 : p11 - p21, da = far02 ? p20 - p00 : p10 - p20;
                const float ba = far02 ? p00 : p20, oa = far02 ? p01 : p21;
                float c2 = na / da * ((float)d0r - ba) + oa;
                if (c2 == c2) {
                    c2 = fminf(fmaxf(c2, -4.0f), (float)is + 4.0f);
                    const int lim = (dir > 0) ? (int)ceilf(c2) : (int)floorf(c2);
                    rfrom[1] = max(min(a_in, lim), 0);
                    rto[1] = min(max(a_in, lim), is - 1);
                }
            }
            // slices of the lines' source arrays: [lo, lo + nb) for both sweeps; only lanes with a sweep touch memory
            const bool on[2] = {act0 && rfrom[0] <= rto[0], act1 && rfrom[1] <= rto[1]};
            int lo[2] = {0, 0}, nbp[2] = {0, 0};
            OFF lid[2];
            uint4 rf[2], rt[2];
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                lid[ph] = ((OFF)(ph * 2 + axis) * B + b) * is + d0r;
                rf[ph] = make_uint4(0u, 0u, 0u, 0u);
                rt[ph] = rf[ph];
                if (on[ph]) {
                    rf[ph] = *hm_at<W32>(lrec, lid[ph] * wpl + (rfrom[ph] >> 6));
                    rt[ph] = *hm_at<W32>(lrec, lid[ph] * wpl + (rto[ph] >> 6));
                }
            }
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const unsigned long long wf = rf[ph].x | ((unsigned long long)rf[ph].y << 32);
                const unsigned long long wt = rt[ph].x | ((unsigned long long)rt[ph].y << 32);
                lo[ph] = (int)rf[ph].z + __popcll(wf & ((1ull << (rfrom[ph] & 63)) - 1ull));
                nbp[ph] = on[ph] ? (int)rt[ph].z + __popcll(wt & (~0ull >> (63 - (rto[ph] & 63)))) - lo[ph] : 0;
            }
            const int nb0 = nbp[0], nb1 = nbp[1];
#ifdef SWEEP_STATS
            {
                const unsigned long long c[7] = {__popcll(__ballot(mine)), __popcll(__ballot(geo)), __popcll(__ballot(act0)),
                                                 __popcll(__ballot(act1)), __popcll(__ballot(nb0 > 0)), __popcll(__ballot(nb1 > 0)),
                                                 (unsigned long long)hm_wave_scan_incl(nb0 + nb1)};
                if (lane == 63) { for (int k = 0; k < 7; ++k) atomicAdd(&g_sweep_n[k], c[k]); atomicAdd(&g_sweep_n[7], 1ull); }
            }
#endif
            // ---- the pairs of the 64 items, flattened over the wave
            const int n = nb0 + nb1;
            const int incl = hm_wave_scan_incl(n);
            const int npairs = __builtin_amdgcn_readlane(incl, 63);
            if (npairs > 0) {
                __builtin_amdgcn_wave_barrier();
                s_start[wv][lane] = incl - n;
                SweepItem it;
                it.x = d1_cross;
                it.c0 = sweep_item_scale(c0, inv_is, pow2, is);
                it.c1 = sweep_item_scale(c1, inv_is, pow2, is);
                it.base0 = (int)(lid[0] * is) + lo[0];
                it.base1 = (int)(lid[1] * is) + lo[1];
                it.nb0 = nb0;
                it.fn = fn;
                {
                    const int m0 = var ? 2 - edge : edge, k1 = edge == 2 ? 0 : edge + 1, m1 = var ? 2 - k1 : k1;
                    const int comp = axis ? 0 : 1;         // row sweeps move x, column sweeps move y
                    it.meta = el | ((2 * m0 + comp) << 4) | ((2 * m1 + comp) << 7) | (use0 ? 1 << 10 : 0) |
                              (use1 ? 1 << 11 : 0) | (nb0 << 12);        // nb0 <= 1024 rides in the upper bits
                }
                s_item[wv][lane].it = it;
                wave_sync();
                const int* st = s_start[wv];
                int cur = -1;
                double acc0 = 0.0, acc1 = 0.0;
                int carry = 0;                 // (item + 1) that owns the pairs running into the current batch
                // 256 pairs per round, four CONSECUTIVE pairs per lane.  Which item a pair belongs to comes from a
                // scatter + max-scan instead of a search: every item whose first pair falls into the round drops its
                // number at that position, and a running maximum carries it over the item's pairs.
#pragma unroll 1
                for (int base = 0; base < npairs; base += 256) {
                    SWEEP_HOOK_PAIR_ROUND(eps, base);
#ifdef SWEEP_STATS
                    if (lane == 0) atomicAdd(&g_sweep_n[11], 1ull);
#endif
#ifdef SWEEP_UNIT_PROFILE
                    ++up_rounds;
#endif
                    int4* hd = reinterpret_cast<int4*>(s_head[wv]);
                    hd[lane] = make_int4(0, 0, 0, 0);
                    wave_sync();
                    const int ex = incl - n;
                    if (n > 0 && ex >= base && ex < base + 256) s_head[wv][ex - base] = lane + 1;
                    wave_sync();
                    const int4 h = hd[lane];
                    int m[4];
                    m[0] = h.x; m[1] = max(m[0], h.y); m[2] = max(m[1], h.z); m[3] = max(m[2], h.w);
                    int inc = m[3];            // inclusive max-scan over the lanes (item numbers are positive: 0 is neutral)
                    inc = max(inc, __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, false));    // row_shr:1
                    inc = max(inc, __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, false));    // row_shr:2
                    inc = max(inc, __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, false));    // row_shr:4
                    inc = max(inc, __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, false));    // row_shr:8
                    inc = max(inc, __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xa, 0xf, false));    // row_bcast:15
                    inc = max(inc, __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xc, 0xf, false));    // row_bcast:31
                    int before = __builtin_amdgcn_update_dpp(0, inc, 0x138, 0xf, 0xf, false);       // wave_shr:1
                    before = max(before, carry);
                    carry = max(carry, __builtin_amdgcn_readlane(inc, 63));
                    // (the LDS pipe is this kernel's busiest unit: a pair reads its item's meta word - with the outward
                    //  count in its upper bits - and one base for the address, then x, c0, c1; base1 / fn only for the rare
                    //  inward pairs)
                    SweepSrc sc[4];
                    int qi[4], qmeta[4];
                    bool ph1[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int p = min(base + 4 * lane + k, npairs - 1);
                        const int i = max(max(before, m[k]) - 1, 0);
                        const int r = max(p - st[i], 0);
                        const SweepItem& qq = s_item[wv][i].it;
                        qmeta[k] = qq.meta;
                        const int q_nb0 = qmeta[k] >> 12;
                        ph1[k] = r >= q_nb0;
                        qi[k] = i;
                        OFF at = (OFF)qq.base0 + r;
                        if (ph1[k]) at = (OFF)qq.base1 + (r - q_nb0);
                        sc[k] = *hm_at<W32>(srcs, at);
                    }
#if SWEEP_COMBINE_MIN <= 64
                    if (base > 0) {     // (wave-uniform: behind a full round, whose accumulators all 64 lanes still carry)
                        // between the loads and their use: the lanes whose first pair of this round starts a new key
                        const int key0 = base + 4 * lane < npairs ? (qmeta[0] & 0x3ff) : cur;
                        sweep_combine_rows(cur, acc0, acc1, cur >= 0 && key0 != cur, lane);
                    }
#endif
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        {
                            // (a slot past the last pair repeats the last pair's loads and is masked: same key, zero term)
                            const bool valid = base + 4 * lane + k < npairs;
                            const SweepItem& qq = s_item[wv][qi[k]].it;
                            const int meta = qmeta[k], key = valid ? (meta & 0x3ff) : cur;
                            if (key != cur) {
                                if (cur >= 0) {
#ifdef SWEEP_HOOK_FLUSH
                                    SWEEP_HOOK_FLUSH(eps, s_fg[wv], lane, cur, acc0, acc1);
#else
                                    double* f = s_fg[wv][cur & 15];
                                    unsafeAtomicAdd(f + ((cur >> 4) & 7), acc0);
                                    unsafeAtomicAdd(f + ((cur >> 7) & 7), acc1);
#endif
                                }
                                cur = key;
                                acc0 = 0.0;
                                acc1 = 0.0;
                            }
                            // (an inward pair counts only if this winding owns the source: masked by a zero `diff`)
                            const bool take = valid && (!ph1[k] || sc[k].owner == qq.fn);
                            sweep_term(take ? (ph1[k] ? sc[k].g : -sc[k].g) : 0.0f, sc[k].d1, qq.x, qq.c0, qq.c1,
                                       (meta & (1 << 10)) != 0, (meta & (1 << 11)) != 0, eps, magic, acc0, acc1);
                        }
                    }
                }
#if SWEEP_COMBINE_MIN <= 64
                if (npairs >= SWEEP_COMBINE_PAIRS) sweep_combine_rows(cur, acc0, acc1, cur >= 0, lane);
#endif
                if (cur >= 0) {
                    double* f = s_fg[wv][cur & 15];
                    unsafeAtomicAdd(f + ((cur >> 4) & 7), acc0);
                    unsafeAtomicAdd(f + ((cur >> 7) & 7), acc1);
                }
            }
            wave_sync();
            }   // stage-2 trips
            // ---- results of the faces of this pass.  The sums are exact (see hm_quant), so a face cut by unit boundaries is
            // simply added by its units with hardware double atomics onto the target the compaction zeroed: any order gives
            // the same value, nobody waits, no partial records, no tickets.
            if (lane < nfp) {
                const SweepFace& ff = s_face[wv][lane].f;
                const int off = ff.off, nit = (int)ff.cum[11];
                const int u_lo = off >> SWEEP_USHIFT, u_hi = (off + nit - 1) >> SWEEP_USHIFT;
                double v[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) v[k] = s_fg[wv][lane][k];
                double* out = parts + (long)ff.bf * 6;
                if (u_lo == u_hi) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) out[k] = v[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 6; ++k) unsafeAtomicAdd(out + k, v[k]);
                }
            }
            if (nfp < SWEEP_PASS_FACES) break;          // the face behind this pass starts beyond the unit
        }   // face passes
#ifdef SWEEP_UNIT_PROFILE
        if (lane == 0 && u < 65536) {
            g_unit_prof[u][0] = (int)(wall_clock64() - up_t0); g_unit_prof[u][1] = up_pass;
            g_unit_prof[u][2] = up_trips | (up_q << 8); g_unit_prof[u][3] = up_rounds;
        }
#endif
    }   // units
    if (ts_on) hm_ts_store(ts_slots, (long)blockIdx.x * 4 + (threadIdx.x >> 6), 1, (unsigned long long)wall_clock64());
}

// ---------------------------------------------------------------- backward, pass 3: vertex gather + projection backward
// adjacency: CSR over vertices, items = face*3 + corner (shared topology) ; grad_verts (B,V,3) overwritten.
__global__ void k_bwd_gather(const double* __restrict__ parts, const int* __restrict__ adj_off,
                             const int* __restrict__ adj_items, const float* __restrict__ verts,
                             const float* __restrict__ K, int B, int V, int F, float orig_size,
                             float* __restrict__ grad_ndc, float* __restrict__ grad_verts)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * V) return;
    const int b = (int)(i / V), v = (int)(i % V);
    double su = 0.0, sv = 0.0;                          // exact: the per-corner sums are multiples of the quantum
    const double2* pf = reinterpret_cast<const double2*>(parts + (long)b * F * 6);
    const int a1 = adj_off[v + 1];
    for (int a = adj_off[v]; a < a1; a += 8) {          // eight corners at a time: item loads, then gradient loads
        int item[8];                                    // face * 3 + corner = double2 index into parts
#pragma unroll
        for (int k = 0; k < 8; ++k) item[k] = a + k < a1 ? adj_items[a + k] : -1;
        double2 g2[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) g2[k] = item[k] >= 0 ? pf[item[k]] : make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < 8; ++k) { su += g2[k].x; sv += g2[k].y; }
    }
    const float gu = (float)su, gv = (float)sv;
    if (grad_ndc) { grad_ndc[3 * i] = gu; grad_ndc[3 * i + 1] = gv; grad_ndc[3 * i + 2] = 0.f; }
    const float* k = K + b * 9;
    const float x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
    const float zz = z + 1e-9f;
    const float du0 = gu * (2.0f / orig_size), dv0 = -gv * (2.0f / orig_size);
    const float dxn = k[0] * du0 + k[3] * dv0;
    const float dyn = k[1] * du0 + k[4] * dv0;
    grad_verts[3 * i] = dxn / zz;
    grad_verts[3 * i + 1] = dyn / zz;
    grad_verts[3 * i + 2] = -(dxn * x + dyn * y) / (zz * zz);
}

// ---------------------------------------------------------------- launchers
void hm_launch_sweep(const SilWs& w, int B, int F, int S, float eps, int sum_log2q, hipStream_t stream)
{
    const RasterTune& tune = hm_raster_tune();
    const int nsort = tune.raster_reorder ? 1 : 0;      // (see hm_tune_raster_reorder: one workgroup more, eight workers less)
    const int blocks = max(8, (min(min(hm_cdiv((long)B * F, 2), tune.sweep_blocks), TS_SWEEP_WGS - 8) & ~7) - 8 * nsort);   // workers: a multiple of 8, see the unit loop
    // (32-bit byte offsets while the largest array the sweep indexes - the per-line source arrays - stays below 4 GB)
    const bool w32 = hm_offsets_fit_32(B, S);
    hipLaunchKernelGGL(w32 ? k_bwd_sweep<true> : k_bwd_sweep<false>, dim3(nsort + blocks), dim3(256), 0, stream, w.sweep,
                       w.idx_map, w.srcs, w.lrec, B, F, S, eps, hm_sum_magic(sum_log2q), w.parts, w.lsum, w.alpha16, w.counter + 24,
                       w.ts + 2 * (ts_raster_units(B, S) + ts_lines_units(B, F, S)), nsort, w.wo_dyn, w.wo_tmp, w.wg_cost,
                       w.counter + 26, (int)ts_raster_units(B, S));
}
void hm_launch_bwd_gather(const SilWs& w, const int* adj_off, const int* adj_items, const float* verts, const float* K, int B,
                          int V, int F, float orig_size, float* grad_ndc, float* grad_verts, hipStream_t stream)
{
    hipLaunchKernelGGL(k_bwd_gather, dim3(hm_cdiv((long)B * V, 256)), dim3(256), 0, stream, w.parts, adj_off, adj_items, verts,
                       K, B, V, F, orig_size, grad_ndc, grad_verts);
}
int hm_sweep_occupancy(int* blocks_per_cu)
{
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, k_bwd_sweep<true>, 256, 0) == hipSuccess ? HM_OK : HM_ERR_LAUNCH;
}
#ifdef SWEEP_UNIT_PROFILE
extern "C" int hm_debug_unit_profile(int* out)              // 65536 x 4 ints, then cleared
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_unit_prof), sizeof(int) * 65536 * 4);
    static int z[65536][4];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_unit_prof), z, sizeof(z));
    return HM_OK;
}
#endif
#ifdef SWEEP_STATS
extern "C" int hm_debug_sweep_stats(unsigned long long* out)
{
    unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sweep_n), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sweep_n), z, sizeof(z));
    return HM_OK;
}
#endif
