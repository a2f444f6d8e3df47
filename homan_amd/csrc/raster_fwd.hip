// raster_fwd.hip -- forward raster (k_raster_fwd), loss reduction, flat-shaded rgb output
#include "raster_ws.h"

// ---------------------------------------------------------------- forward raster
// Workgroup = 4 wavefronts = one 32x32-sample region (a 2x2 block of 8x8-pixel output tiles) of one frame.
//   1. binning on the fly: the workgroup scans the 8-byte screen boxes of the frame (coalesced) and keeps the
//      (face, winding) entries overlapping its region in an LDS candidate list;
//   2. one thread per candidate builds the face record (edge vectors, reciprocal depths, barycentric inverse) in LDS
//      and counts the 4x4-sample blocks of (face box & region);
//   3. the (candidate, block) units of the pass are FLATTENED over the 256 threads (unit u -> candidate by a binary
//      search of the exclusive unit counts): every thread tests the 16 samples of its block against its face and
//      resolves visibility with ds_min_u64 on an LDS z-buffer keyed (depth bits << 32 | face index) -- the smaller
//      depth wins and equal depths go to the smaller index, which is the strict-z-test-in-ascending-face-order rule
//      of the rasteriser.  Faces are ~50 samples large: broadcasting one face to a 256-sample tile (round-1 design)
//      spent >90 % of the sample tests outside the face's box;
//   4. epilogue: one wave per 8x8 output tile reads its samples back (lane = output pixel, 2x2 samples).
// Outputs: idx_map (B,is,is) int32; alpha16 (B,is,is/16) u16 bit-plane; pooled (B,S,S);
// optional fused loss terms: dimg = keep*(keep*pool-ref), partials (B,ntiles,4); optional pooled depth.
#ifdef RASTER_TRACE
// per-workgroup phase durations of wave 0 in wall-clock ticks (10 ns), stored once at the workgroup's end - no atomics, nothing
// that would stretch the kernel being measured (the RASTER_PHASES counters below add global atomics per candidate: fine for
// COUNTS, useless for times).  [blockIdx][0..5] = scan, records (+ barrier), own near units, barrier + hidden-block depths, far
// units (+ barrier), epilogue; [6] = start tick (low 32 bits), [7] = candidates near | far << 16, [8] = units near | far << 16
#define RASTER_TRACE_WGS 65536
__device__ unsigned g_raster_trace[RASTER_TRACE_WGS][9];
__device__ int g_raster_trace_F = 0, g_raster_trace_depth = -1;     // record only launches over meshes of F faces (0: any) / with (1) or without (0) a depth output
#define RPH_MARK(k) do { if (tid == 0) { const unsigned long long t_ = wall_clock64(); rtr[k] += (unsigned)(t_ - rtr_t); rtr_t = t_; } } while (0)
#elif defined(RASTER_PHASES)
__device__ unsigned long long g_raster_ph[24];   // cycles of wave 0: scan, near records, near units, far hz + records, far units, tail; workgroups: active, idle; units near / far
#define RPH_MARK(k) do { if (tid == 0) { const unsigned long long t_ = clock64(); rph[k] += t_ - rph_t; rph_t = t_; } } while (0)
#else
#define RPH_MARK(k)
#endif
#ifndef CAND_CAP
#define CAND_CAP 512       // faces scanned per binning round (<= 2 entries each)
#endif
#ifndef RB_PASS
#define RB_PASS 128        // candidates per record pass (<= threads; a region sees ~60 candidates, and LDS is occupancy)
#endif
#ifndef HM_PRUNE
#define HM_PRUNE 1
#endif
#ifndef RASTER_WPE
#define RASTER_WPE 6       // waves per SIMD the register budget is sized for (LDS: 25 KB per workgroup = 6 per CU)
#endif
// The body of the forward raster: `blk` = index of the workgroup WITHIN its render (k_raster_fwd: blockIdx.x; k_raster_fwd_multi:
// blockIdx.x minus the first workgroup of the render it serves).  Every argument is workgroup-uniform.
__device__ __forceinline__ void raster_fwd_body(
    const int blk, const float* __restrict__ faces9, const FaceBox* __restrict__ boxes, int B, int F, int S, float znear,
    float zfar, int* __restrict__ idx_map, unsigned short* __restrict__ alpha16, float* __restrict__ pooled,
    const float* __restrict__ keep, const float* __restrict__ ref, float* __restrict__ dimg,
    float* __restrict__ partials, const int* __restrict__ work_order, unsigned char* __restrict__ owned,
    float* __restrict__ pooled_depth, unsigned short* __restrict__ planes,
    int* __restrict__ bin_cnt, const int* __restrict__ bin_list, unsigned int* __restrict__ done, int reset_bins,
    unsigned char* __restrict__ region_state, int persistent, float* __restrict__ alpha_full, int mask_shared,
    float* __restrict__ dimg_full, const unsigned int* __restrict__ hint, unsigned long long* __restrict__ ts_slots,
    int* __restrict__ wo_dyn, unsigned int* __restrict__ wg_cost)
{
    const unsigned long long ts_t0 = (unsigned long long)wall_clock64();       // (see hm_ts_enabled)
    __shared__ unsigned long long zb[32 * 32];
    __shared__ int cand[2 * CAND_CAP];
    __shared__ float4 recs[RB_PASS][5];
    __shared__ int ustart[RB_PASS];
    __shared__ unsigned czn[RB_PASS];           // per candidate: bits of the nearest depth it can produce
    __shared__ unsigned hz[64];                 // per 4x4 block: largest owner depth (bits) of its 16 samples
    __shared__ unsigned short uq[RASTER_WAVES][128];   // per wave: far-class units that passed the hidden-block test
    __shared__ int wsum[RASTER_WAVES];
    __shared__ int cand_n[2];
    // (wave-uniform values the compiler cannot prove uniform - the wave index, the work-order entry - go through
    //  readfirstlane: everything derived from them (region box, corner coordinates, bin, frame offsets) then lives in scalar
    //  registers and is computed once per wave by the scalar unit instead of per lane)
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, tid = threadIdx.x;
    const int is = 2 * S, tiles_x = S / HM_TILE, ntiles = tiles_x * tiles_x, regions_x = tiles_x / 2;
    // dispatch order = work_order[block] = (frame << 16 | region): expensive (frame, region) pairs first so that the
    // cheap ones fill the tail of the launch (default: centre of the ROI outwards; calibrated: by candidate count)
    // Adaptive order (hm_tune_raster_reorder, wo_dyn != NULL): every workgroup leaves the time it took in wg_cost and the
    // entry it served in wo_dyn; a workgroup of the backward's sweep launch sorts the entries by that time, longest first,
    // for the NEXT forward of this workspace (hint word 2 then says "use wo_dyn").  What is expensive moves during a fit;
    // an order taken from the poses at its start is stale after a few dozen iterations.
    const bool dyn = wo_dyn && hint[2] != 0u;
    const int wo = __builtin_amdgcn_readfirstlane(dyn ? wo_dyn[blk]
                                                     : work_order ? work_order[blk] : (int)(((blk % B) << 16) | (blk / B)));
    if (wo_dyn && !dyn && threadIdx.x == 0) wo_dyn[blk] = wo;
    const int region = wo & 0xffff, b = wo >> 16;
    const int rx = region % regions_x, ry = region / regions_x;
    const int tx = 2 * rx + (w & 1), ty = 2 * ry + (w >> 1);
    const int tile = ty * tiles_x + tx;

    // sample box of the workgroup's region
    const int gx0 = rx * 2 * HM_STILE, gx1 = gx0 + 2 * HM_STILE - 1;
    const int gy1 = is - 1 - ry * 2 * HM_STILE, gy0 = gy1 - (2 * HM_STILE - 1);
    const float gcx0 = (float)(2 * gx0 + 1 - is) / (float)is, gcx1 = (float)(2 * gx1 + 1 - is) / (float)is;
    const float gcy0 = (float)(2 * gy0 + 1 - is) / (float)is, gcy1 = (float)(2 * gy1 + 1 - is) / (float)is;
    const bool pow2 = (is & (is - 1)) == 0;
    const float inv_is = 1.0f / (float)is;       // exact for powers of two
    const unsigned long long zb_empty = ((unsigned long long)__float_as_uint(zfar) << 32) | 0xffffffffull;

    int had_any = 0;       // block-uniform: some face overlaps this region
    const uint2* bx = reinterpret_cast<const uint2*>(boxes) + (long)b * F;
    // faces to scan: the bin of this region's super-region (or the whole frame without bins)
    const int nsx = (is + (1 << hm_sr_shift(is)) - 1) >> hm_sr_shift(is);
    const int sr = (gy0 >> hm_sr_shift(is)) * nsx + (gx0 >> hm_sr_shift(is));
    // the two words every workgroup needs first, requested together: the size of its bin and the state of its outputs
    unsigned char* rstate = region_state + (long)b * regions_x * regions_x + region;
    const int nscan = bin_cnt ? bin_cnt[b * nsx * nsx + sr] : F;
    const int rstate0 = persistent ? (int)*rstate : 0;
    const int* scan = bin_cnt ? bin_list + ((long)b * nsx * nsx + sr) * F : nullptr;
    // An empty bin in front of outputs that already hold the empty pattern: nothing to rasterise, nothing to write (~60 %
    // of the workgroups of a clip, every iteration) - leave before touching LDS.  (The bin ticket still has to be drawn.)
    const bool idle = nscan == 0 && rstate0 == 1;
    const bool ts_on = hm_ts_enabled(hint) && tid == 0;
    if (ts_on) hm_ts_store(ts_slots, blk, 0, ts_t0);
#ifdef RASTER_TRACE
    unsigned rtr[6] = {0, 0, 0, 0, 0, 0}, rtr_cand = 0, rtr_units = 0;
    unsigned long long rtr_t = ts_t0;
#endif
#ifdef RASTER_PHASES
    unsigned long long rph[6] = {0, 0, 0, 0, 0, 0}, rph_t = clock64();
    unsigned long long rph_units[3] = {0, 0, 0};
    unsigned rph_pairs = 0;      // per lane: covered samples
    __shared__ unsigned s_rph_iters;      // wave trips of the covered-sample loop, all waves
    if (tid == 0) s_rph_iters = 0u;
    __syncthreads();
#endif
    if (!idle) {
#pragma unroll
        for (int k = 0; k < 4; ++k) zb[tid + 256 * k] = zb_empty;
    }
    // winding class rasterised FIRST (scheduling hint, see hm_sil_hint_near_winding): on a closed mesh one winding class
    // holds the camera-facing surface and owns every sample, the other is hidden behind it
    const int near_w = hint ? (int)(hint[0] & 1u) : 0;
    // ---- one (candidate, 4x4 block) unit: inside tests, hidden-sample pruning, depth + z-buffer min
    auto unit_body = [&](const int i, const int k) {
        const float4 r0 = recs[i][0], r1 = recs[i][1], r4 = recs[i][4];
        const int fn = __float_as_int(r4.z), pk = __float_as_int(r4.w);
        const int nbx = (pk >> 6) & 15;
        const int kby = (k * (pk >> 10)) >> 16;          // k / nbx (k < 64, reciprocal packed by the record builder)
        const int sx0 = 4 * ((pk & 7) + (k - kby * nbx)), sy0 = 4 * (((pk >> 3) & 7) + kby);
        // edge by edge (row / column terms of one edge live at a time: the register budget is the kernel's occupancy):
        // sample (j, c4) is inside iff for every edge !(row term < column term)
        float Xs[4], Ys[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xi = gx0 + sx0 + j, yi = gy0 + sy0 + j;
            // sample positions: (2 i + 1 - is) / is ; a power-of-two `is` makes the product with 1/is the same float as
            // the IEEE quotient (eight 14-instruction divisions per unit otherwise)
            const float xn = (float)(2 * xi + 1 - is), yn = (float)(2 * yi + 1 - is);
            Xs[j] = pow2 ? xn * inv_is : xn / (float)is;
            Ys[j] = pow2 ? yn * inv_is : yn / (float)is;
        }
        // "rv < cv" as the SIGN BIT of rv - cv, shifted into the mask by one v_alignbit: two instructions per (sample, edge)
        // where the compare needed v_cmp + s_nop + v_cndmask + v_or.  Same decision as the compare for every pair of FINITE
        // operands once rv cannot be -0 (the one case where the signs lie: (-0) - (+0) = -0, while -0 < +0 is false): rv + 0.0f
        // turns -0 into +0 and nothing else, four additions per edge.  Distinct floats never difference to zero, equal ones
        // give +0.  Faces whose products could overflow were culled by the face setup (|NDC| <= 1e15, like the oracle).
        // The three edges' differences of a sample are OR-ed (the sign bit of the OR is "outside some edge") into sixteen
        // accumulators, one edge at a time - the empty asm keeps the compiler from holding all 48 differences for one v_or3
        // per sample, which cost the kernel its register budget (18 spilled values) - then one v_alignbit per sample.
        unsigned acc[16];
        {
            const float vx[3] = {r0.x, r0.z, r1.x}, vy[3] = {r0.y, r0.w, r1.y};
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const int e1 = e == 2 ? 0 : e + 1;
                const float ex = vx[e1] - vx[e], ey = vy[e1] - vy[e];
                float rv[4], cv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { rv[j] = (Ys[j] - vy[e]) * ex + 0.0f; cv[j] = (Xs[j] - vx[e]) * ey; }
#pragma unroll
                for (int idx = 0; idx < 16; ++idx) {
                    const unsigned d = __float_as_uint(rv[idx >> 2] - cv[idx & 3]);
                    acc[idx] = e ? (acc[idx] | d) : d;
                }
#pragma unroll
                for (int idx = 0; idx < 16; ++idx) asm volatile("" : "+v"(acc[idx]));
            }
        }
        unsigned outside = 0u;
#pragma unroll
        for (int idx = 15; idx >= 0; --idx) outside = __builtin_amdgcn_alignbit(outside, acc[idx], 31);      // sample idx -> bit idx
        unsigned inside = ~outside & 0xffffu;
        if (inside == 0u) return;
#if HM_PRUNE
        // samples already owned by something nearer than the nearest point of this face cannot change (the
        // interpolated depth is a weighted harmonic mean of the vertex depths; 1e-5 covers its rounding)
        {
            const unsigned zn = czn[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (((inside >> (4 * j)) & 0xfu) == 0u) continue;
                const uint4 ka = *reinterpret_cast<const uint4*>(&zb[(sy0 + j) * 32 + sx0]);
                const uint4 kb = *reinterpret_cast<const uint4*>(&zb[(sy0 + j) * 32 + sx0 + 2]);
                unsigned keepm = (zn > ka.y ? 0u : 1u) | (zn > ka.w ? 0u : 2u) | (zn > kb.y ? 0u : 4u) | (zn > kb.w ? 0u : 8u);
                inside &= ~(0xfu << (4 * j)) | (keepm << (4 * j));
            }
            if (inside == 0u) return;
        }
#endif
#ifdef RASTER_PHASES
        rph_pairs += __popc(inside);
        {   // trips of the loop below = the fullest of the lanes that got here
            const unsigned mx = (unsigned)hm_wave_max((float)__popc(inside));
            if (lane == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&s_rph_iters, mx);
        }
#endif
        const float4 r2 = recs[i][2], r3 = recs[i][3];
        const float rz0 = r1.z, rz1 = r1.w, rz2 = r2.x;
        const float iv[9] = {r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w, r4.x, r4.y};
        const int xb = gx0 + sx0, yb = gy0 + sy0;
        while (inside) {
            const int sidx = __ffs((int)inside) - 1;
            inside &= inside - 1;
            const int j = sidx >> 2, c4 = sidx & 3;
            const float xf = (float)(xb + c4), yf = (float)(yb + j);
            float wgt[3], ws = 0.0f;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float t = iv[3 * q] * xf;
                t = t + iv[3 * q + 1] * yf;
                t = t + iv[3 * q + 2];
                t = fminf(fmaxf(t, 0.0f), 1.0f);
                wgt[q] = t;
                ws += t;
            }
            float sum = wgt[0] * rz0;
            sum = sum + wgt[1] * rz1;
            sum = sum + wgt[2] * rz2;
            const float zp = ws / sum;
            if (zp > znear && zp < zfar)
                atomicMin(&zb[(sy0 + j) * 32 + sx0 + c4],
                          ((unsigned long long)__float_as_uint(zp) << 32) | (unsigned)fn);
        }
    };
    for (int cbase = 0; cbase < nscan; cbase += CAND_CAP) {
        if (tid == 0) { cand_n[0] = 0; cand_n[1] = 0; }
        __syncthreads();
        uint2 v[CAND_CAP / 256];
        int vf[CAND_CAP / 256];
#pragma unroll
        for (int k = 0; k < CAND_CAP / 256; ++k) {
            const int e = cbase + k * 256 + tid;
            vf[k] = e < nscan ? (scan ? scan[e] : e) : -1;
        }
#pragma unroll
        for (int k = 0; k < CAND_CAP / 256; ++k) v[k] = vf[k] >= 0 ? bx[vf[k]] : make_uint2(0u, 0u);
#pragma unroll
        for (int k = 0; k < CAND_CAP / 256; ++k) {
            const int fi = vf[k];
            const int x0 = v[k].x & 0x3fff, y0 = (int)(v[k].x >> 16), x1 = (int)(v[k].y & 0xffff), y1 = (int)(v[k].y >> 16);
            const unsigned m = (x1 < gx0 || x0 > gx1 || y1 < gy0 || y0 > gy1) ? 0u : ((v[k].x >> 14) & 3u);
#pragma unroll
            for (int var = 0; var < 2; ++var) {
                const bool hit = (m >> var) & 1u;
                const unsigned long long bal = __ballot(hit);
                if (bal == 0ull) continue;
                // the near class fills the candidate array from the front, the far class from the back
                const int cls = var == near_w ? 0 : 1;
                int basep = 0;
                if (lane == 0) basep = atomicAdd(&cand_n[cls], __popcll(bal));
                basep = __builtin_amdgcn_readfirstlane(basep);
                const int at = basep + __popcll(bal & ((1ull << lane) - 1ull));
                if (hit) cand[cls ? 2 * CAND_CAP - 1 - at : at] = fi | (var << 30);
            }
        }
        __syncthreads();
        RPH_MARK(0);
        had_any |= cand_n[0] | cand_n[1];
        // Records of BOTH winding classes in one pass: wave 0 builds up to 64 near-class records, wave 1 up to 64 far-class
        // records at the same time (a region sees ~27 + ~31 candidates) - one round trip to the packed faces and one stretch of
        // record arithmetic per round instead of one per class (the far class's pass was a quarter of the workgroup's time,
        // with three waves waiting at its barrier).  Then the near units, the hidden-block depths they leave, the far units.
        const int n_near = cand_n[0], n_far = cand_n[1];
        for (int e0 = 0; e0 < n_near || e0 < n_far; e0 += RB_PASS / 2) {
            // ---- one thread per candidate: face record + number of 4x4 blocks of (box & region)
            int units = 0;
            const int cls = tid >> 6, ci = e0 + (tid & 63);
            if (tid < RB_PASS && ci < (cls ? n_far : n_near)) {
                const int e = cand[cls ? 2 * CAND_CAP - 1 - ci : ci];
                const int fi = e & 0x3fffffff, var = e >> 30;
                const float* src = faces9 + ((long)b * F + fi) * 9;
                const uint2 u = bx[fi];          // (requested with the vertices: one round trip per record, not two)
                float f[9];
                if (var == 0) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) f[k] = src[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { f[3 * k] = src[3 * (2 - k)]; f[3 * k + 1] = src[3 * (2 - k) + 1]; f[3 * k + 2] = src[3 * (2 - k) + 2]; }
                }
                float p[3][2];
#pragma unroll
                for (int k = 0; k < 3; ++k) { p[k][0] = topix(f[3 * k], is); p[k][1] = topix(f[3 * k + 1], is); }
                const float inv[9] = {
                    p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
                    p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
                    p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
                const float den = p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) +
                                  p[1][0] * (p[2][1] - p[0][1]);
                // conservative reject: some edge has all four region corners outside by more than the rounding noise
                // of the edge function (the function is affine, so its extremes over the region sit at the corners)
                bool miss = (den == 0.0f);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int k1 = (k + 1) % 3;
                    const float ax = f[3 * k], ay = f[3 * k + 1], ex = f[3 * k1] - ax, ey = f[3 * k1 + 1] - ay;
                    bool all_out = true;
                    float mag = 0.f;
                    float lhs[4], rhs[4];
#pragma unroll
                    for (int cnr = 0; cnr < 4; ++cnr) {
                        const float X = (cnr & 1) ? gcx1 : gcx0, Y = (cnr & 2) ? gcy1 : gcy0;
                        lhs[cnr] = (Y - ay) * ex;
                        rhs[cnr] = (X - ax) * ey;
                        mag = fmaxf(mag, fabsf(lhs[cnr]) + fabsf(rhs[cnr]));
                    }
#pragma unroll
                    for (int cnr = 0; cnr < 4; ++cnr) all_out = all_out && (lhs[cnr] < rhs[cnr] - 1e-5f * mag);
                    miss = miss || all_out;
                }
                if (!miss) {
                    const int x0 = u.x & 0x3fff, y0 = (int)(u.x >> 16), x1 = (int)(u.y & 0xffff), y1 = (int)(u.y >> 16);
                    // region-local 4x4 block range
                    const int bx0 = (max(x0, gx0) - gx0) >> 2, bx1 = (min(x1, gx1) - gx0) >> 2;
                    const int by0 = (max(y0, gy0) - gy0) >> 2, by1 = (min(y1, gy1) - gy0) >> 2;
                    const int nbx = bx1 - bx0 + 1;
                    units = nbx * (by1 - by0 + 1);
                    const float rz0 = 1.0f / f[2], rz1 = 1.0f / f[5], rz2 = 1.0f / f[8];
                    recs[tid][0] = make_float4(f[0], f[1], f[3], f[4]);
                    recs[tid][1] = make_float4(f[6], f[7], rz0, rz1);
                    recs[tid][2] = make_float4(rz2, inv[0] / den, inv[1] / den, inv[2] / den);
                    recs[tid][3] = make_float4(inv[3] / den, inv[4] / den, inv[5] / den, inv[6] / den);
                    recs[tid][4] = make_float4(inv[7] / den, inv[8] / den, __int_as_float(fi + var * F),
                                               __int_as_float(bx0 | (by0 << 3) | (nbx << 6) | (((0x10000 + nbx - 1) / nbx) << 10)));
                    // nearest depth the face can produce (the interpolated depth is a weighted harmonic mean of the
                    // vertex depths), with 1e-5 of slack for its rounding: the pruning threshold of its units
                    czn[tid] = __float_as_uint((1.0f / fmaxf(rz0, fmaxf(rz1, rz2))) * (1.0f - 1e-5f));
                }
            }
            // ---- exclusive prefix of the unit counts, per class = per wave (waves 0 / 1; no cross-wave sums)
            const int incl = hm_wave_scan_incl(units);
            if (lane == 63 && w < 2) wsum[w] = incl;
            if (tid < RB_PASS) ustart[tid] = incl - units;
            __syncthreads();
            const int total_near = wsum[0], total_far = wsum[1];
#ifdef RASTER_TRACE
            rtr_cand += (unsigned)min(n_near - e0, RB_PASS / 2) * (e0 < n_near) + ((unsigned)min(n_far - e0, RB_PASS / 2) * (e0 < n_far) << 16);
            rtr_units += (unsigned)total_near + ((unsigned)total_far << 16);
#endif
#ifdef RASTER_PHASES
            if (tid < RB_PASS && ci < (cls ? n_far : n_near)) atomicAdd(&g_raster_ph[12 + cls], 1ull);
            rph_units[0] += total_near;
            rph_units[1] += total_far;
#endif
            RPH_MARK(1);
            // ---- near class: flattened (candidate, block) units, one per thread and trip
            for (int u = tid; u < total_near; u += 256) {
                int i = 0;
#pragma unroll
                for (int stp = RB_PASS / 4; stp > 0; stp >>= 1)
                    if (ustart[i + stp] <= u) i += stp;             // (i + stp <= 63)
                unit_body(i, u - ustart[i]);
            }
            RPH_MARK(2);
            if (total_far > 0) {
                // hidden-block test for the far class: per 4x4 block, the largest owner depth the near class (and earlier
                // rounds) left in the z-buffer (zfar while any of its samples is empty)
                if (tid < 64) hz[tid] = 0u;
                __syncthreads();
                {
                    const int blk = tid >> 2, row = (blk >> 3) * 4 + (tid & 3), col0 = (blk & 7) * 4;
                    const uint4 ka = *reinterpret_cast<const uint4*>(&zb[row * 32 + col0]);
                    const uint4 kb = *reinterpret_cast<const uint4*>(&zb[row * 32 + col0 + 2]);
                    atomicMax(&hz[blk], max(max(ka.y, ka.w), max(kb.y, kb.w)));
                }
                __syncthreads();
                RPH_MARK(3);
                // ---- far class: most units sit behind the near surface.  A wave first tests 64 units against the
                // hidden-block depths (one LDS word each) and queues the survivors; the unit body runs on full waves
                // of survivors only (a divergent early-out would leave the wave paying for its one visible unit)
                int qn = 0;
                for (int u0 = 64 * w; u0 < total_far; u0 += 256) {
                    const int u = u0 + lane;
                    bool pass = false;
                    int ent = 0;
                    if (u < total_far) {
                        int i = 0;
#pragma unroll
                        for (int stp = RB_PASS / 4; stp > 0; stp >>= 1)
                            if (ustart[RB_PASS / 2 + i + stp] <= u) i += stp;
                        i += RB_PASS / 2;
                        const int k = u - ustart[i];
                        const int pk = __float_as_int(recs[i][4].w);
                        const int nbx = (pk >> 6) & 15, kby = (k * (pk >> 10)) >> 16;
                        const int blk = (((pk >> 3) & 7) + kby) * 8 + (pk & 7) + (k - kby * nbx);
                        pass = !(czn[i] > hz[blk]);
                        ent = i | (k << 8);
                    }
                    const unsigned long long bal = __ballot(pass);
#ifdef RASTER_PHASES
                    if (lane == 0) atomicAdd(&g_raster_ph[21], (unsigned long long)__popcll(bal));
#endif
                    if (pass) uq[w][qn + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)ent;
                    qn += __popcll(bal);
                    wave_sync();
                    if (qn >= 64) {
                        qn -= 64;
                        const int e = uq[w][qn + lane];
                        wave_sync();
                        unit_body(e & 0xff, e >> 8);
                    }
                }
                if (lane < qn) {
                    const int e = uq[w][lane];
                    unit_body(e & 0xff, e >> 8);
                }
            }
            __syncthreads();
            RPH_MARK(4);
        }
    }
    __syncthreads();
    // the last of the (4 or 16) workgroups that read a bin empties it for the next forward.  One ticket word per bin:
    // returning atomics on a single word from all 7680 workgroups serialise (measured +43 us on the launch).  (Drawing the
    // ticket right after the scan and using the answer at the very end was measured: k_raster_fwd 51.8 -> 55.5 us.)
    if (tid == 0 && bin_cnt && reset_bins) {
        const int per_side = (1 << hm_sr_shift(is)) / (2 * HM_STILE);
        const int rw = min(per_side, regions_x - per_side * (gx0 >> hm_sr_shift(is))), rh = min(per_side, regions_x - per_side * (gy0 >> hm_sr_shift(is)));
        const int slot = b * nsx * nsx + sr;
        if (atomicAdd(done + slot, 1u) == (unsigned)(rw * rh) - 1u) {
            bin_cnt[slot] = 0;
            atomicExch(done + slot, 0u);
        }
    }

    // An empty region whose outputs already hold the empty pattern has nothing to write: ~60 % of the regions of a clip
    // are background in every iteration, and their epilogues (loads of the loss inputs, ~6 KB of stores) were a quarter
    // of the kernel.  Only valid when the caller keeps passing the same output / loss-input buffers (`persistent`).
    if (persistent && !had_any && rstate0 == 1) {
#ifdef RASTER_PHASES
        if (tid == 0) atomicAdd(&g_raster_ph[7], 1ull);
#endif
        if (wg_cost && tid == 0) wg_cost[blk] = 0u;
        return;
    }
    __syncthreads();          // every thread has read the state before thread 0 rewrites it below
    if (tid == 0) *rstate = (persistent && !had_any) ? 1 : 0;

    // ---- epilogue: this lane's output pixel and its 2x2 samples (flip: output row r <-> sample rows is-1-2r-dy)
    const int r = ty * HM_TILE + (lane >> 3), c = tx * HM_TILE + (lane & 7);
    const int xi0 = 2 * c, yi0 = is - 1 - 2 * r;   // sample (dy,dx): yi = yi0 - dy, xi = xi0 + dx
    float zmin[4];
    int imin[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int dy = s4 >> 1, dx = s4 & 1;
        const unsigned long long key = zb[(yi0 - dy - gy0) * 32 + (xi0 + dx - gx0)];
        zmin[s4] = __uint_as_float((unsigned)(key >> 32));
        imin[s4] = (int)(unsigned)(key & 0xffffffffull);      // 0xffffffff -> -1
    }

    // ---- outputs
    int* im = idx_map + (long)b * is * is;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        hm_out_store2(im + (long)(yi0 - dy) * is + xi0, imin[2 * dy], imin[2 * dy + 1]);
    }
    // faces that own a sample (benign same-value races).  One-byte stores are partial-line writes that never merge
    // across the XCDs' L2s, so a sample is flagged only by the first lane of its run: not if the same face owns the
    // previous sample of this pixel, the same sample of the previous lane (left neighbour) or of the lane eight back
    // (upper neighbour) -- the lowest lane holding a face always stores.
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int left = __builtin_amdgcn_update_dpp(-2, imin[s], 0x138, 0xf, 0xf, false);            // wave_shr:1
        const int up = __builtin_amdgcn_ds_bpermute(((lane - 8) & 63) << 2, imin[s]);
        if (imin[s] >= 0 && (s == 0 || imin[s] != imin[s - 1]) && imin[s] != left && (lane < 8 || imin[s] != up))
            owned[(long)b * 2 * F + imin[s]] = 1;
    }
    // alpha bit-plane: 16 sample rows x 16 bits for this tile
    const Ballots4 bal = {__ballot(imin[0] >= 0), __ballot(imin[1] >= 0), __ballot(imin[2] >= 0), __ballot(imin[3] >= 0)};
    if (lane < 16) {
        const int rr = lane >> 1, dy = lane & 1;       // tile-local output row, sub-row
        const unsigned long long be = dy ? bal.s2 : bal.s0, bo = dy ? bal.s3 : bal.s1;    // (selects: see emit_planes)
        const unsigned a = (unsigned)(be >> (8 * rr)) & 0xffu;      // dx = 0 -> even bits
        const unsigned o = (unsigned)(bo >> (8 * rr)) & 0xffu;      // dx = 1 -> odd bits
        const unsigned word = spread8(a) | (spread8(o) << 1);
        alpha16[(((long)b * (is / 16) + ty) * (is / 16) + tx) * 16 + lane] = (unsigned short)word;    // tile-blocked: 32 B / tile
    }
    const int cnt = (imin[0] >= 0) + (imin[1] >= 0) + (imin[2] >= 0) + (imin[3] >= 0);
    const float pool = 0.25f * (float)cnt;
    const long po = ((long)b * S + r) * S + c;
    hm_out_store(pooled + po, pool);
    // anti_aliasing=False rendering: the silhouette is the sample grid itself (flipped), no pooling
    if (alpha_full) {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
            *reinterpret_cast<float2*>(alpha_full + ((long)b * is + 2 * r + dy) * is + xi0) =
                make_float2(imin[2 * dy] >= 0 ? 1.f : 0.f, imin[2 * dy + 1] >= 0 ? 1.f : 0.f);
    }
    // depth image of nr.Renderer.render (homan.py:391,406): z-buffer (far where empty), flipped, 2x2 average pooled
    if (pooled_depth) pooled_depth[po] = (((zmin[0] + zmin[1]) + zmin[2]) + zmin[3]) / 4.0f;
    if (partials && dimg_full) {
        // per-SAMPLE masked L2 (rendering without anti-aliasing, reference homan/pose_optimization.py:140-143): keep / ref
        // are (is,is) images [shared by all frames when mask_shared], dimg_full = keep * (keep * alpha - ref) per sample
        const float* kb = keep + ((mask_shared & 1) ? 0 : (long)b * is * is);
        const float* rb = ref + ((mask_shared & 1) ? 0 : (long)b * is * is);
        const bool ps_store = !(mask_shared & 2);          // bit 1: no per-sample outputs (the backward runs in mode 5)
        float sqs = 0.f, ins = 0.f, uns = 0.f;
        unsigned long long nq[4], pq[4];     // constant indices only (unrolled)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const long at = (long)(2 * r + dy) * is + xi0;
            const float2 k2 = *reinterpret_cast<const float2*>(kb + at), r2 = *reinterpret_cast<const float2*>(rb + at);
            const float i0 = k2.x * (imin[2 * dy] >= 0 ? 1.f : 0.f), i1 = k2.y * (imin[2 * dy + 1] >= 0 ? 1.f : 0.f);
            const float d0 = i0 - r2.x, d1 = i1 - r2.y;
            const float g0 = k2.x * d0, g1 = k2.y * d1;
            if (ps_store) *reinterpret_cast<float2*>(dimg_full + (long)b * is * is + at) = make_float2(g0, g1);
            sqs += d0 * d0 + d1 * d1;
            ins += i0 * r2.x + i1 * r2.y;
            uns += fminf(fmaxf(i0 + r2.x, 0.0f), 1.0f) + fminf(fmaxf(i1 + r2.y, 0.0f), 1.0f);
            nq[2 * dy] = __ballot(g0 < 0.0f); pq[2 * dy] = __ballot(g0 > 0.0f);
            nq[2 * dy + 1] = __ballot(g1 < 0.0f); pq[2 * dy + 1] = __ballot(g1 > 0.0f);
        }
        const Ballots4 nbv = {nq[0], nq[1], nq[2], nq[3]}, pbv = {pq[0], pq[1], pq[2], pq[3]};
        emit_planes(bal, nbv, pbv, b, B, is, tx, ty, lane, planes);
        const float sq = hm_wave_sum(sqs), inter = hm_wave_sum(ins), uni = hm_wave_sum(uns);
        if (lane == 0) {
            float* o = partials + ((long)b * ntiles + tile) * 4;
            o[0] = sq; o[1] = inter; o[2] = uni; o[3] = 0.f;
        }
    } else if (partials) {
        const long pm = (mask_shared & 1) ? (long)r * S + c : po;
        const float kp = keep[pm], rf = ref[pm];      // (requesting them at kernel start was measured: no gain, +3 registers)
        const float image = kp * pool;
        const float diff = image - rf;
        hm_out_store(dimg + po, kp * diff);
        // sweep planes of the backward for a positive upstream gradient (sign(g) = sign(dimg)), see k_bwd_masks
        {
            const unsigned long long nb1 = __ballot(kp * diff < 0.0f), pb1 = __ballot(kp * diff > 0.0f);
            const Ballots4 nbv = {nb1, nb1, nb1, nb1}, pbv = {pb1, pb1, pb1, pb1};
            emit_planes(bal, nbv, pbv, b, B, is, tx, ty, lane, planes);
        }
        const float sq = hm_wave_sum(diff * diff);
        const float inter = hm_wave_sum(image * rf);
        const float uni = hm_wave_sum(fminf(fmaxf(image + rf, 0.0f), 1.0f));
        if (lane == 0) {
            float* o = partials + ((long)b * ntiles + tile) * 4;
            o[0] = sq; o[1] = inter; o[2] = uni; o[3] = 0.f;
        }
    }
    if (ts_on) hm_ts_store(ts_slots, blk, 1, (unsigned long long)wall_clock64());
    if (wg_cost && tid == 0) wg_cost[blk] = (unsigned)min((unsigned long long)wall_clock64() - ts_t0, 0xfffffffeull) + 1u;
#ifdef RASTER_TRACE
    RPH_MARK(5);
    if (tid == 0 && blk < RASTER_TRACE_WGS && (g_raster_trace_F == 0 || g_raster_trace_F == F) &&
        (g_raster_trace_depth < 0 || g_raster_trace_depth == (pooled_depth ? 1 : 0))) {
        unsigned* o = g_raster_trace[blk];
        for (int k = 0; k < 6; ++k) o[k] = rtr[k];
        o[6] = (unsigned)ts_t0; o[7] = rtr_cand; o[8] = rtr_units;
    }
#endif
#ifdef RASTER_PHASES
    RPH_MARK(5);
    if (tid == 0) {
        for (int k = 0; k < 6; ++k) atomicAdd(&g_raster_ph[k], rph[k]);
        atomicAdd(&g_raster_ph[6], 1ull);
        atomicAdd(&g_raster_ph[8], rph_units[0]);
        atomicAdd(&g_raster_ph[9], rph_units[1]);
    }
    {
        const unsigned wp = (unsigned)hm_wave_sum((float)rph_pairs);
        if (lane == 0) atomicAdd(&g_raster_ph[11], (unsigned long long)wp);
        __syncthreads();
        if (tid == 0) atomicAdd(&g_raster_ph[10], (unsigned long long)s_rph_iters);
    }
#endif
}

__global__ __launch_bounds__(64 * RASTER_WAVES) __attribute__((amdgpu_waves_per_eu(RASTER_WPE, 8))) void k_raster_fwd(RasterFwdK a)
{
    HM_CHAIN_KERNEL();
    raster_fwd_body((int)blockIdx.x, a.faces9, a.boxes, a.B, a.F, a.S, a.znear, a.zfar, a.idx_map, a.alpha16, a.pooled, a.keep, a.ref,
                    a.dimg, a.partials, a.work_order, a.owned, a.pooled_depth, a.planes, a.bin_cnt, a.bin_list, a.done, a.reset_bins,
                    a.region_state, a.persistent, a.alpha_full, a.mask_shared, a.dimg_full, a.hint, a.ts_slots, a.wo_dyn, a.wg_cost);
}
// Several renders (their own meshes, cameras, outputs, workspaces) as ONE launch: the workgroups of render g are
// [first[g], first[g + 1]); which render a workgroup serves is a scalar search over <= HM_MAX_RENDERS thresholds, its arguments
// are scalar loads from the kernarg segment as before.  Same body, same results as one launch per render.
__global__ __launch_bounds__(64 * RASTER_WAVES) __attribute__((amdgpu_waves_per_eu(RASTER_WPE, 8))) void k_raster_fwd_multi(RasterFwdMulti m)
{
    HM_CHAIN_KERNEL();
    int g = 0;
#pragma unroll
    for (int k = 1; k < HM_MAX_RENDERS; ++k) g += (k < m.n && (int)blockIdx.x >= m.first[k]) ? 1 : 0;
    g = __builtin_amdgcn_readfirstlane(g);
    const RasterFwdK& a = m.r[g];
    raster_fwd_body((int)blockIdx.x - m.first[g], a.faces9, a.boxes, a.B, a.F, a.S, a.znear, a.zfar, a.idx_map, a.alpha16, a.pooled, a.keep,
                    a.ref, a.dimg, a.partials, a.work_order, a.owned, a.pooled_depth, a.planes, a.bin_cnt, a.bin_list, a.done,
                    a.reset_bins, a.region_state, a.persistent, a.alpha_full, a.mask_shared, a.dimg_full, a.hint, a.ts_slots,
                    a.wo_dyn, a.wg_cost);
}


__global__ __launch_bounds__(256) void k_sil_reduce(const float* __restrict__ partials, int B, int ntiles,
                                                     const float* __restrict__ keep_sum, float* __restrict__ frame_rec,
                                                     float* __restrict__ out, float* __restrict__ frame_out, int clip_len,
                                                     int out_stride)
{
    HM_LATENCY_KERNEL();
    sil_reduce_frame(blockIdx.x, partials, ntiles, keep_sum, frame_rec, out, frame_out, clip_len, out_stride);
}

// ---------------------------------------------------------------- rgb output (nr `render`: lighting + texture_size 1)
// Flat-shaded colour image of the index map left by the last forward: per output pixel the 2x2 samples read their owner
// face's single texel times light = ambient + directional * relu(<n, dir>), n = normalize((v0-v1) x (v2-v1), eps 1e-5)
// of the owner WINDING in camera space (the reversed copy of fill_back flips n); empty samples read the background;
// vertical flip and 2x2 average as for the other outputs.  grid (S*S/256, B).  rgb (B,3,S,S).
__global__ __launch_bounds__(256) void k_shade_rgb(const int* __restrict__ idx_map, const float* __restrict__ verts,
                                                   const int* __restrict__ faces, int faces_bstride,
                                                   const float* __restrict__ textures, int B, int V, int F, int S,
                                                   float dx, float dy, float dz, float amb, float dirw, float bg0,
                                                   float bg1, float bg2, float* __restrict__ rgb)
{
    const int b = blockIdx.y, pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= S * S) return;
    const int r = pix / S, c = pix - r * S, is = 2 * S;
    const int* idx = idx_map + (long)b * is * is;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int yi = is - 1 - (2 * r + (q >> 1)), xi = 2 * c + (q & 1);
        const int fn = idx[(long)yi * is + xi];
        float col[3] = {bg0, bg1, bg2};
        if (fn >= 0) {
            const int f = fn >= F ? fn - F : fn;
            const int* fc = faces + (long)b * faces_bstride + 3 * f;
            const float* v0 = verts + ((long)b * V + fc[0]) * 3;
            const float* v1 = verts + ((long)b * V + fc[1]) * 3;
            const float* v2 = verts + ((long)b * V + fc[2]) * 3;
            const float a[3] = {v0[0] - v1[0], v0[1] - v1[1], v0[2] - v1[2]};
            const float e[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
            float n[3] = {a[1] * e[2] - a[2] * e[1], a[2] * e[0] - a[0] * e[2], a[0] * e[1] - a[1] * e[0]};
            const float len = fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-5f);
            float cs = (n[0] / len) * dx + (n[1] / len) * dy + (n[2] / len) * dz;
            if (fn >= F) cs = -cs;
            const float light = amb + dirw * fmaxf(cs, 0.f);
            const float* t = textures + ((long)b * F + f) * 3;
            col[0] = t[0] * light; col[1] = t[1] * light; col[2] = t[2] * light;
        }
        acc[0] += col[0]; acc[1] += col[1]; acc[2] += col[2];
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) rgb[(((long)b * 3 + ch) * S + r) * S + c] = 0.25f * acc[ch];
}

// ---------------------------------------------------------------- launchers
static RasterFwdK raster_fwd_k(const SilWs& w, const RasterFwdArgs& a)
{
    RasterFwdK k = {w.faces9, w.boxes, a.B, a.F, a.S, a.znear, a.zfar, w.idx_map, w.alpha16, a.pooled, a.keep, a.ref, w.dimg,
                    a.fused ? w.partials : (float*)nullptr, a.work_order, w.owned, a.pooled_depth, w.planes, a.bins,
                    w.bin_list, w.bin_done, a.reset_bins, w.region_state, a.persistent, a.alpha_full, a.mask_shared,
                    a.per_sample_grad ? w.gimg : (float*)nullptr, w.counter + 24, w.ts, a.reorder ? w.wo_dyn : (int*)nullptr,
                    a.reorder ? w.wg_cost : (unsigned int*)nullptr};
    return k;
}
void hm_launch_raster_fwd(const SilWs& w, const RasterFwdArgs& a, hipStream_t stream)
{
    const int ntiles = (a.S / 8) * (a.S / 8);
    hipLaunchKernelGGL(k_raster_fwd, dim3(a.B * (ntiles / RASTER_WAVES)), dim3(64 * RASTER_WAVES), a.lds_pad, stream,
                       raster_fwd_k(w, a));
}
void hm_launch_raster_fwd_multi(const SilWs* w, const RasterFwdArgs* a, int n, hipStream_t stream)
{
    RasterFwdMulti m;
    m.n = n;
    int at = 0;
    for (int g = 0; g < HM_MAX_RENDERS; ++g) {
        const int q = g < n ? g : n - 1;           // (unused slots repeat the last render: never selected)
        m.r[g] = raster_fwd_k(w[q], a[q]);
        m.first[g] = at;
        if (g < n) at += a[g].B * ((a[g].S / 8) * (a[g].S / 8) / RASTER_WAVES);
    }
    m.first[HM_MAX_RENDERS] = at;
    hipLaunchKernelGGL(k_raster_fwd_multi, dim3(at), dim3(64 * RASTER_WAVES), a[0].lds_pad, stream, m);
}
void hm_launch_sil_reduce(const SilWs& w, int B, int S, const float* keep_sum, float* loss_out, float* frame_out, int clip_len,
                          int out_stride, hipStream_t stream)
{
    hipLaunchKernelGGL(k_sil_reduce, dim3(B), dim3(256), 0, stream, w.partials, B, (S / 8) * (S / 8), keep_sum, w.frame_rec,
                       loss_out, frame_out, clip_len, out_stride);
}
void hm_launch_shade_rgb(const SilWs& w, const float* verts, const int* faces, int faces_bstride, const float* textures, int B,
                         int V, int F, int S, const float* light_dir, float amb, float dirw, const float* background, float* rgb,
                         hipStream_t stream)
{
    hipLaunchKernelGGL(k_shade_rgb, dim3(hm_cdiv((long)S * S, 256), B), dim3(256), 0, stream, w.idx_map, verts, faces,
                       faces_bstride, textures, B, V, F, S, light_dir[0], light_dir[1], light_dir[2], amb, dirw, background[0],
                       background[1], background[2], rgb);
}
int hm_raster_fwd_occupancy(int* blocks_per_cu)
{
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, k_raster_fwd, 64 * RASTER_WAVES, 0) == hipSuccess ? HM_OK
                                                                                                                        : HM_ERR_LAUNCH;
}
#ifdef RASTER_TRACE
extern "C" int hm_debug_raster_trace_filter(int F, int depth)
{
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_raster_trace_F), &F, sizeof(int));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_raster_trace_depth), &depth, sizeof(int));
    return HM_OK;
}
extern "C" int hm_debug_raster_trace(unsigned* out, int nwg)          // (nwg, 9) -> HOST buffer; then cleared
{
    (void)hipDeviceSynchronize();
    const size_t bytes = sizeof(unsigned) * 9 * (size_t)min(nwg, RASTER_TRACE_WGS);
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_raster_trace), bytes);
    void* dev = nullptr;
    (void)hipGetSymbolAddress(&dev, HIP_SYMBOL(g_raster_trace));
    (void)hipMemset(dev, 0, sizeof(unsigned) * 9 * RASTER_TRACE_WGS);
    return HM_OK;
}
#endif
#ifdef RASTER_PHASES
extern "C" int hm_debug_raster_phases(unsigned long long* out)
{
    unsigned long long z[24] = {0};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_raster_ph), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_raster_ph), z, sizeof(z));
    return HM_OK;
}
#endif
