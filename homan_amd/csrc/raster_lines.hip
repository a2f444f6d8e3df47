// raster_lines.hip -- backward, pass 1 / 2a: sample-gradient masks, line expansion, work list of the edge sweeps
#include "raster_ws.h"

// ---------------------------------------------------------------- backward, pass 1: masks + sample-gradient image
// g(b,r,c) = dL/dpooled.  mode 0: gin is that image.  mode 1: gin is dimg (keep*(keep*pool-ref)) and
// g = upstream[0] * 2 * dimg / keep_sum / B (the fused masked-MSE of losses.py:188-194).
// Emits gimg (B,S,S) and row/column bit masks, two planes each: plane 0 = samples with alpha==0 and g<0 ("wants to be
// filled", walked by the outward sweeps), plane 1 = samples with alpha==1 and g>0 ("wants to be emptied", the only
// samples the inward sweeps can collect from).
__global__ __launch_bounds__(256) void k_bwd_masks(const float* __restrict__ gin, int mode,
                                                   const float* __restrict__ upstream,
                                                   const float* __restrict__ keep_sum, int B, int S,
                                                   const unsigned short* __restrict__ alpha16,
                                                   float* __restrict__ gimg, unsigned short* __restrict__ planes,
                                                   int clip_len)
{
    // fused loss with a positive upstream gradient: the forward raster already emitted these planes (sign(g) = sign(dimg))
    // and k_bwd_lines derives g from dimg itself
    if (mode == 1 && upstream[0] > 0.0f) return;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int is = 2 * S, tiles_x = S / HM_TILE, ntiles = tiles_x * tiles_x;
    const int tile = blockIdx.x * 4 + w, b = blockIdx.y;
    if (tile >= ntiles) return;
    const int ty = tile / tiles_x, tx = tile % tiles_x;
    const int rr = lane >> 3, cc = lane & 7;
    const int r = ty * HM_TILE + rr, c = tx * HM_TILE + cc;
    const long po = ((long)b * S + r) * S + c;
    // alpha bits of this lane's 4 samples
    unsigned long long cq[4];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const unsigned aw = alpha16[(((long)b * (is / 16) + ty) * (is / 16) + tx) * 16 + 2 * rr + dy];
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) cq[2 * dy + dx] = __ballot((aw >> (2 * cc + dx)) & 1u);
    }
    unsigned long long nq[4], pq[4];
    if (mode == 3) {
        // anti_aliasing=False: the image IS the sample grid (vertically flipped); gin / gimg are (B,is,is)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const long at = ((long)b * is + 2 * r + dy) * is + 2 * c;
            const float2 g2 = *reinterpret_cast<const float2*>(gin + at);
            *reinterpret_cast<float2*>(gimg + at) = g2;
            nq[2 * dy] = __ballot(g2.x < 0.0f); pq[2 * dy] = __ballot(g2.x > 0.0f);
            nq[2 * dy + 1] = __ballot(g2.y < 0.0f); pq[2 * dy + 1] = __ballot(g2.y > 0.0f);
        }
    } else {
        float g = gin[po];
        if (mode == 1) {
            float s = upstream[0] * 2.0f;
            g = s * g / keep_sum[b / clip_len] / (float)clip_len;
        }
        gimg[po] = g;
        const unsigned long long nb1 = __ballot(g < 0.0f), pb1 = __ballot(g > 0.0f);
#pragma unroll
        for (int k = 0; k < 4; ++k) { nq[k] = nb1; pq[k] = pb1; }
    }
    const Ballots4 cov = {cq[0], cq[1], cq[2], cq[3]}, nbv = {nq[0], nq[1], nq[2], nq[3]}, pbv = {pq[0], pq[1], pq[2], pq[3]};
    emit_planes(cov, nbv, pbv, b, B, is, tx, ty, lane, planes);
}

// ---------------------------------------------------------------- backward, pass 2b (edge sweeps): work list
// The work of the NMR pseudo-gradient is the set of (face winding, edge, axis, d0) ITEMS - one per sample line an edge
// crosses - and, under every item, the (item, source) PAIRS of its two sweeps.  Both levels are flattened:
//  * compaction blocks riding at the front of the k_bwd_lines launch turn the faces that own at least one sample into a
//    table of 64-byte records {pixel-space corners, cumulative item counts of the 12 (winding, edge, axis) families}
//    laid end to end in one global item space (block scan + one 64-bit atomic per block for the block's base: table
//    order == item order).  Faces that own nothing get their zero gradient written there and never reach the sweep.
//  * a UNIT is 64 consecutive items = one wave-iteration of k_bwd_sweep, whichever faces they belong to (typically
//    1-3; a large face spreads over several units, i.e. over several waves).  `ufirst[u]` names the face holding
//    item 64u.  Every lane rebuilds its item from its face's record in LDS; nothing is wave-serial.
//  * an item resolves its two sweeps to slices of per-line source arrays (k_bwd_lines); the pairs of the 64 items are
//    flattened over the wave: pair p goes to lane p % 64, which finds its item by a binary search of the items'
//    exclusive pair counts in LDS (pairs per item are heavy-tailed: mean 5, lines tangent to the band hold hundreds).
//  * a lane keeps running sums while its pairs stay on one (face, corner) target and flushes them into the face's six
//    LDS accumulators when the target changes.  A face inside one unit is stored directly; a face cut by one unit
//    boundary is added by its two units with hardware float atomics onto a zeroed target (commutative: deterministic);
//    a face spread over three or more units leaves per-unit partials and the unit that draws the last ticket adds them
//    in unit order (deterministic).
// parts (B,F,3 mesh corners,2): d/d(x, y) of the NDC face vertices.

// item count of the (edge, axis) line family between end points with sweep-axis coordinates a0, a1
__device__ __forceinline__ int sweep_family(float a0, float a1, int is, int& d0_from)
{
    d0_from = 0;
    if (!(a0 != a1)) return 0;
    d0_from = (int)fmaxf(ceilf(fminf(a0, a1)), 0.0f);
    const int d0_to = (int)fminf(fmaxf(a0, a1), (float)is - 1.0f);
    return max(0, d0_to - d0_from + 1);
}

// record of face slot bf (corners, cumulative family counts, ids) -> its item count (0: owns no sample / culled)
__device__ __forceinline__ int sweep_face_record(long bf, const float* __restrict__ faces9, const FaceBox* __restrict__ boxes,
                                                 const unsigned char* __restrict__ owned, int F, int is, SweepFace& rec)
{
    const int b = (int)(bf / F), fi = (int)(bf - (long)b * F);
    // (box, flags and corners are requested together: one round trip for the block, whether or not the face is active)
    const unsigned mask = (reinterpret_cast<const uint2*>(boxes)[bf].x >> 14) & 3u;
    const unsigned char own0 = owned[(long)b * 2 * F + fi], own1 = owned[(long)b * 2 * F + F + fi];
    const float* src = faces9 + bf * 9;
    float sx[3], sy[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { sx[k] = src[3 * k]; sy[k] = src[3 * k + 1]; }
    // a winding that owns no sample has no in-pixel of its own and nothing to sweep inwards over: zero gradient
    const bool act0 = (mask & 1u) && own0;
    const bool act1 = (mask & 2u) && own1;
    if (!(act0 || act1)) return 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) { rec.px[k] = topix(sx[k], is); rec.py[k] = topix(sy[k], is); }
    int n = 0;
#pragma unroll
    for (int var = 0; var < 2; ++var)
#pragma unroll
        for (int e = 0; e < 3; ++e)
#pragma unroll
            for (int axis = 0; axis < 2; ++axis) {
                const int k0 = e, k1 = (e + 1) % 3;
                const int v0 = var ? 2 - k0 : k0, v1 = var ? 2 - k1 : k1;
                int from;
                const int c = sweep_family(axis ? rec.py[v0] : rec.px[v0], axis ? rec.py[v1] : rec.px[v1], is, from);
                if (var ? act1 : act0) n += c;
                rec.cum[var * 6 + e * 2 + axis] = (unsigned short)n;
            }
    rec.bf = (int)bf;
    rec.b = b;
    return n;
}

// one block = 256 * fpt consecutive face slots, fpt (1..4) per thread in thread-major order: one face per thread keeps
// the blocks' dependent chain short (a clip: a few hundred blocks), four amortise it and the same-address atomics when
// there are thousands of blocks (500 candidate poses).  `nblk` = number of compaction blocks of the launch.  Two passes
// over the block's faces: item counts -> block scan -> one atomic for the block's base -> records (rebuilt rather than
// kept: the line expansion shares this kernel and its register budget).
// A launch over several clips (clip_len frames each) gives every clip its own run of `nblk / clips` compaction blocks over
// its own face slots: a block never straddles two clips, so the unit composition of a clip - and with it every summation
// order of its gradients - is the one of a single-clip launch.
__device__ __forceinline__ void sweep_compact(int blk, int nblk, int fpt, const float* __restrict__ faces9,
                                              const FaceBox* __restrict__ boxes, const unsigned char* __restrict__ owned,
                                              int B, int F, int is, double* __restrict__ parts, const SweepList& sl,
                                              int clip_len)
{
    __shared__ int s_wsum[4][2];
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per_clip = nblk / (B / clip_len), clip = blk / per_clip;
    const long cbase = (long)clip * clip_len * F;
    const long bf0 = cbase + ((long)(blk - clip * per_clip) * 256 + tid) * fpt, nbf = cbase + (long)clip_len * F;
    int nk[4], n = 0, hasf = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        SweepFace rec;
        nk[k] = (k < fpt && bf0 + k < nbf) ? sweep_face_record(bf0 + k, faces9, boxes, owned, F, is, rec) : 0;
        n += nk[k];
        hasf += nk[k] > 0 ? 1 : 0;
    }
    // block-exclusive scan of (items, faces)
    const int inc_i = hm_wave_scan_incl(n), inc_f = hm_wave_scan_incl(hasf);
    if (lane == 63) { s_wsum[wv][0] = inc_i; s_wsum[wv][1] = inc_f; }
    __syncthreads();
    int pre_i = 0, pre_f = 0, tot_i = 0, tot_f = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < wv) { pre_i += s_wsum[k][0]; pre_f += s_wsum[k][1]; }
        tot_i += s_wsum[k][0];
        tot_f += s_wsum[k][1];
    }
    // A block's items start on a unit boundary (its total is padded to a multiple of SWEEP_UNIT): which items share a unit, and
    // so the order in which every sum below is formed, depends only on the block's own faces - never on the order in
    // which the blocks drew their bases.  Items in the padding belong to no face.
    if (tid == 0)
        s_base = tot_f ? atomicAdd(sl.cnt, ((unsigned long long)tot_f << 32) | (unsigned)((tot_i + SWEEP_UNIT - 1) & ~(SWEEP_UNIT - 1))) : 0ull;
    __syncthreads();
    const unsigned long long base = s_base;
    int off = (int)(base & 0xffffffffull) + pre_i + inc_i - n;
    int idx = (int)(base >> 32) + pre_f + inc_f - hasf;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long bf = bf0 + k;
        if (k >= fpt || bf >= nbf) break;
        bool zero = nk[k] == 0;
        if (nk[k] > 0) {
            SweepFace rec;
            const int nn = sweep_face_record(bf, faces9, boxes, owned, F, is, rec);
            const int u_lo = off >> SWEEP_USHIFT, u_hi = (off + nn - 1) >> SWEEP_USHIFT;
            const bool over = u_hi >= sl.ucap || u_hi + idx >= sl.slot_cap;
            rec.off = off;
            rec.flags = over ? 1 : 0;
            zero = u_hi > u_lo;                                  // accumulated with double atomics by its units
            const uint4* r4 = reinterpret_cast<const uint4*>(&rec);
            uint4* t4 = reinterpret_cast<uint4*>(sl.tab + idx);
#pragma unroll
            for (int q = 0; q < 4; ++q) t4[q] = r4[q];
            sl.offs[idx] = off;
            for (int u = (off + SWEEP_UNIT - 1) >> SWEEP_USHIFT; (u << SWEEP_USHIFT) < off + nn && u < sl.ucap; ++u)
                sl.ufirst[u] = (unsigned)idx;
            off += nn;
            ++idx;
        }
        if (zero) {
            double* o = parts + bf * 6;
#pragma unroll
            for (int q = 0; q < 6; ++q) o[q] = 0.0;
        }
    }
    // the last block publishes the totals and re-arms the counters for the next launch.  (Only the counts travel
    // through this ticket - thread 0 consumed the return value of its own add above -; the records are read by the next
    // kernel, so nobody waits for their stores here.)
    if (tid == 0) {
        const unsigned int t = atomicAdd(sl.done, 1u);
        if (t == (unsigned)nblk - 1u) {
            sl.total[0] = atomicExch(sl.cnt, 0ull);
            atomicExch(sl.done, 0u);
        }
    }
}

// ---------------------------------------------------------------- backward, pass 2a: per-line source lists
// Each sweep of an (edge, axis, d0) item collects from the set bits of ONE line of a plane, restricted to a range.
// The lines are shared by all the items that cross them (~140 per line), so they are expanded once: a wave per
// (plane, axis, frame, line) turns the bit line into a compact, position-sorted array of sources
// {d1, sample gradient, owner face} plus the cumulative bit count at every 64-bit word.  An item then knows its
// sources as the contiguous slice [lo, lo+nb) of that array (two popcounts), with no bit walking and no dependent
// gradient / owner loads.

// 16 lanes per line (one DPP row) and LINES_NL consecutive lines per row, 16 rows per workgroup.  (History: a wave per line spent
// its life waiting on three dependent memory round trips with 8 of 64 lanes loading; four lines per wave quartered the waves in
// flight; two lines per ROW halve the workgroups again - 2 300 instead of 4 200 at one clip, about one resident round - and
// the two lines' mask words arrive in the same 4-byte loads: consecutive lines are neighbouring 16-bit words of the same
// tiles.  The sources of a 64-sample word are requested together, not one dependent load per set bit.  Four lines per row
// make the rows with a line tangent to a band the launch's tail: 22 us at two, 32 us at four, 25.5 us at one.)
#ifndef LINES_NL
#define LINES_NL 2
#endif
// which 16 * LINES_NL lines the i-th of the n line workgroups takes (scheduling only)
#ifdef LINES_REVERSED
#define LINES_BLK(i, n) ((n) - 1 - (i))
#else
#define LINES_BLK(i, n) (i)
#endif
template <bool W32>
__global__ __launch_bounds__(256) void k_bwd_lines(const unsigned short* __restrict__ planes,
                                                   const float* __restrict__ gimg, const float* __restrict__ dimg,
                                                   int mode, const float* __restrict__ upstream,
                                                   const float* __restrict__ keep_sum,
                                                   const int* __restrict__ idx_map, int B, int S,
                                                   SweepSrc* __restrict__ srcs, uint4* __restrict__ lrec,
                                                   int ncomp, int fpt, const float* __restrict__ faces9,
                                                   const FaceBox* __restrict__ boxes,
                                                   const unsigned char* __restrict__ owned, int F,
                                                   double* __restrict__ parts, SweepList sl, int clip_len,
                                                   unsigned short* __restrict__ lsum, int nred,
                                                   const float* __restrict__ red_partials, float* __restrict__ frame_rec,
                                                   float* __restrict__ loss_out, int out_stride,
                                                   const unsigned int* __restrict__ ts_flag,
                                                   unsigned long long* __restrict__ ts_slots)
{
    HM_CHAIN_KERNEL();
    const unsigned long long ts_t0 = (unsigned long long)wall_clock64();
    const bool ts_on = hm_ts_enabled(ts_flag) && threadIdx.x == 0;
    if (ts_on) hm_ts_store(ts_slots, blockIdx.x, 0, ts_t0);
    const int blk = (int)blockIdx.x;
    __shared__ unsigned long long s_w[16][LINES_NL][SWEEP_CUMW];
    __shared__ int s_ex[16][LINES_NL][SWEEP_CUMW];
    // the first `ncomp` workgroups build the work list of the edge sweeps (independent of the lines: one launch for both)
    if (blk < ncomp) {
        sweep_compact(blk, ncomp, fpt, faces9, boxes, owned, B, F, 2 * S, parts, sl, clip_len);
        if (ts_on) hm_ts_store(ts_slots, blockIdx.x, 1, (unsigned long long)wall_clock64());
        return;
    }
    // the next `nred` (= B or 0) finish the forward's fused loss: one launch less on the chain of a caller that only needs
    // the loss value for its log (see hm_sil_bwd_clips)
    if (blk < ncomp + nred) {
        sil_reduce_frame(blk - ncomp, red_partials, (S / 8) * (S / 8), keep_sum, frame_rec, loss_out, nullptr, clip_len,
                         out_stride);
        if (ts_on) hm_ts_store(ts_slots, blockIdx.x, 1, (unsigned long long)wall_clock64());
        return;
    }
    typedef typename HmOff<W32>::t OFF;
    const int l = threadIdx.x & 15, grp = threadIdx.x >> 4;
    // (the 16 rows of a workgroup leave at different times: each folds its exit into the workgroup's own end slot)
    const bool ts_row = l == 0 && hm_ts_enabled(ts_flag);
    const int is = 2 * S, wpl = is / 64, T = is / 16;
    // The workgroup's 16 * LINES_NL consecutive lines (is is a multiple of 64 >= 16 * LINES_NL: they share plane, orientation
    // and frame), decomposed ONCE per workgroup in scalar registers - L = ((pl * 2 + axis) * B + b) * is + d0 - and the row's
    // first line from there (the 64-bit divisions per lane were a tenth of this kernel's instructions)
    const unsigned Lw = (unsigned)(LINES_BLK(blk - ncomp - nred, (int)gridDim.x - ncomp - nred)) * (16u * LINES_NL);
    const bool valid = Lw < 4u * (unsigned)B * (unsigned)is;
    const unsigned fr = Lw / (unsigned)is;                                  // (plane-orientation, frame) index
    const int d00 = (int)(Lw - fr * (unsigned)is) + grp * LINES_NL, b = (int)(fr % (unsigned)B), pa = (int)(fr / (unsigned)B);
    const OFF L0 = (OFF)Lw + (OFF)(grp * LINES_NL);
    const int axis = pa & 1, pl = pa >> 1;
    // 64 samples [64 l, 64 l + 64) of the LINES_NL lines: four tiles' words, and in every tile the lines' words are neighbours
    // (axis 1: sample row t = is - 1 - d0, so line j is word LINES_NL - 1 - j of the aligned group; axis 0: word j) - see
    // hm_plane_word64
    unsigned long long mine[LINES_NL];
#pragma unroll
    for (int j = 0; j < LINES_NL; ++j) mine[j] = 0ull;
    if (valid && l < wpl) {
        unsigned long long raw[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            OFF at;          // hm_plane_at, in the offset type
            if (axis) {
                const int t = is - 1 - d00 - (LINES_NL - 1);          // lowest sample row of the group
                at = ((((OFF)b * T + (t >> 4)) * T + (4 * l + jj)) * 4 + pl) * 16 + (t & 15);
            } else {
                at = ((((OFF)b * T + (T - 1 - (4 * l + jj))) * T + (d00 >> 4)) * 4 + (2 + pl)) * 16 + (d00 & 15);
            }
#if LINES_NL == 4
            raw[jj] = *reinterpret_cast<const unsigned long long*>(hm_at<W32>(planes, at));
#elif LINES_NL == 2
            raw[jj] = *reinterpret_cast<const unsigned int*>(hm_at<W32>(planes, at));
#else
            raw[jj] = *hm_at<W32>(planes, at);
#endif
        }
#pragma unroll
        for (int j = 0; j < LINES_NL; ++j) {
            const int sel = axis ? LINES_NL - 1 - j : j;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) mine[j] |= ((raw[jj] >> (16 * sel)) & 0xffffull) << (16 * jj);
        }
    }
    bool any = false;
#pragma unroll
    for (int j = 0; j < LINES_NL; ++j) {
        const OFF L = L0 + j;
        const int d0 = d00 + j;
        // exclusive prefix of the word popcounts (wpl <= 16 words: one 16-lane row scan)
        const int c = __popcll(mine[j]);
        int incl = c;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);    // row_shr:1
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);    // row_shr:2
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);    // row_shr:4
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);    // row_shr:8
        const int excl = incl - c;
        // line record: {64 mask bits, number of set bits before them} per word, one 16-byte load for a sweep end point and
        // one cache line per line of up to 512 samples
        if (valid && l < wpl) *hm_at<W32>(lrec, L * wpl + l) = make_uint4((unsigned)mine[j], (unsigned)(mine[j] >> 32), (unsigned)excl, 0u);
        // line summary for the sweeps' early-out, 8 bytes per (line, plane) at lsum[(((b*2 + axis)*is + d0)*2 + pl)*4 ..]:
        // {first set position, last set position + 1 (0: empty line), mask of the non-empty 64-sample words}: an item whose
        // sweep range cannot reach a set bit never looks further
        int lo = mine[j] ? 64 * l + __builtin_ctzll(mine[j]) : 0xffff, hi = mine[j] ? 64 * l + 64 - __builtin_clzll(mine[j]) : 0;
        lo = min(lo, __builtin_amdgcn_update_dpp(0xffff, lo, 0x111, 0xf, 0xf, false)); hi = max(hi, __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, false));
        lo = min(lo, __builtin_amdgcn_update_dpp(0xffff, lo, 0x112, 0xf, 0xf, false)); hi = max(hi, __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, false));
        lo = min(lo, __builtin_amdgcn_update_dpp(0xffff, lo, 0x114, 0xf, 0xf, false)); hi = max(hi, __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, false));
        lo = min(lo, __builtin_amdgcn_update_dpp(0xffff, lo, 0x118, 0xf, 0xf, false)); hi = max(hi, __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, false));
        const unsigned wm = (unsigned)(__ballot(mine[j] != 0ull) >> (16 * (grp & 3))) & 0xffffu;
        if (valid && l == 15)       // row_shr scans: lane 15 of the row holds the row's result
            *reinterpret_cast<uint2*>(hm_at<W32>(lsum, ((((OFF)b * 2 + axis) * is + d0) * 2 + pl) * 4)) =
                make_uint2((unsigned)lo | ((unsigned)hi << 16), wm);
        any = any || wm != 0u;
        s_w[grp][j][l] = mine[j];
        s_ex[grp][j][l] = excl;
    }
    wave_sync();          // (a row's words are written and read by lanes of ONE wave)
    if (!valid || !any) {
        if (ts_row) atomicMax(ts_slots + 2 * blockIdx.x + 1, (unsigned long long)wall_clock64());
        return;
    }
    // fused loss, positive upstream: g = upstream * 2 * dimg / keep_sum / B (the arithmetic of k_bwd_masks), no gimg pass
    const bool from_dimg = mode == 2 || (mode == 1 && upstream[0] > 0.0f);     // (modes 3 / 4 read gimg per sample below)
    const float* gi = (from_dimg ? dimg : gimg) + (long)b * S * S;
    const float gs = from_dimg ? upstream[0] * 2.0f : 0.f, ks = from_dimg ? keep_sum[b / clip_len] : 1.f;
    const float up4 = (mode == 4 || mode == 5) ? upstream[b] * 2.0f : 0.f;
    const int* idx = idx_map + (long)b * is * is;
    const float* gfull = gimg + (long)b * is * is;
#pragma unroll 1
    for (int j = 0; j < LINES_NL; ++j) {
        const int d0 = d00 + j;
        const OFF out0 = (L0 + j) * is;          // the line's source array starts at srcs[out0]
        for (int k = 0; k < wpl; ++k) {
            const unsigned long long w = s_w[grp][j][k];
            if (w == 0ull) continue;
            const int base = s_ex[grp][j][k];
            // the (up to four) sources of this lane in the word: all loads first, then the records
            float gl[4];
            int ow[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pos = 16 * q + l;
                gl[q] = 0.f;
                ow[q] = -1;
                if (!((w >> pos) & 1ull)) continue;
                const int d1 = (k << 6) + pos;
                const int xi = axis ? d1 : d0, yi = axis ? d0 : d1;
                if (mode == 5) gl[q] = pl ? 1.0f : -1.0f;         // binary masks: keep (keep alpha - ref) is -1 where an uncovered
                                                                  // sample pulls and +1 where a covered one pushes - no load
                else if (mode == 3 || mode == 4) gl[q] = gfull[(is - 1 - yi) * is + xi];      // per-sample gradient (no anti-aliasing); (frame-local: < 2^22)
                else gl[q] = gi[((is - 1 - yi) >> 1) * S + (xi >> 1)];
                if (pl) ow[q] = idx[yi * is + xi];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pos = 16 * q + l;
                if (!((w >> pos) & 1ull)) continue;
                SweepSrc r;
                r.d1 = (k << 6) + pos;
                float g = gl[q];
                if (mode == 4 || mode == 5) g = up4 * g;                      // fused per-sample L2
                else if (mode != 3) {
                    if (from_dimg) g = gs * g / ks / (float)clip_len;
                    g = 0.25f * g;
                }
                r.g = g;
                r.owner = ow[q];
                *hm_at<W32>(srcs, out0 + (OFF)(base + __popcll(w & ((1ull << pos) - 1ull)))) = r;
            }
        }
    }
    if (ts_row) atomicMax(ts_slots + 2 * blockIdx.x + 1, (unsigned long long)wall_clock64());
}

// ---------------------------------------------------------------- launchers
void hm_launch_bwd_masks(const SilWs& w, const float* gin, int mode, const float* upstream, const float* keep_sum, int B, int S,
                         int clip_len, hipStream_t stream)
{
    const int ntiles = (S / 8) * (S / 8);
    hipLaunchKernelGGL(k_bwd_masks, dim3(hm_cdiv(ntiles, 4), B), dim3(256), 0, stream, gin, mode, upstream, keep_sum, B, S,
                       w.alpha16, w.gimg, w.planes, clip_len);
}
// pass 2a (+ the work list of pass 2b in its first workgroups, + optionally the forward's loss reduction in front of both)
void hm_launch_lines(const SilWs& w, int B, int F, int S, int mode, const float* upstream, const float* keep_sum, int clip_len,
                     hipStream_t stream, float* loss_out, int out_stride)
{
    const int nred = loss_out ? B : 0;
    // work-list blocks: per clip (see sweep_compact), sized by the clip, so that a clip is cut into the same blocks
    // whether it is launched alone or in a batch
    const int fpt = (long)clip_len * F >= 400000 ? 4 : 1;      // faces per thread of the work-list blocks
    const int ncomp = (B / clip_len) * hm_cdiv((long)clip_len * F, 256 * fpt);
    const bool w32 = hm_offsets_fit_32(B, S);
    hipLaunchKernelGGL(w32 ? k_bwd_lines<true> : k_bwd_lines<false>, dim3(ncomp + nred + hm_cdiv(4L * B * 2 * S, 16 * LINES_NL)), dim3(256), 0, stream, w.planes,
                       w.gimg, w.dimg, mode, upstream, keep_sum, w.idx_map, B, S, w.srcs, w.lrec, ncomp, fpt, w.faces9, w.boxes,
                       w.owned, F, w.parts, w.sweep, clip_len, w.lsum, nred, w.partials, w.frame_rec, loss_out, out_stride,
                       w.counter + 24, w.ts + 2 * ts_raster_units(B, S));
}
