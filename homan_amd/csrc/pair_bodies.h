// pair_bodies.h -- bodies of three small reduction kernels as device functions on VIRTUAL block coordinates (bx, by, gdx
// instead of blockIdx.x / blockIdx.y / gridDim.x), so that each runs either as its own launch (losses.hip: k_smooth, k_inter;
// contact.hip: k_nn_min) or as a block range of the fused pair-terms launch (pairterms.hip).  256 threads per block in
// every case; the arithmetic and the summation orders do not depend on which way a body is launched.
#pragma once
#include "hm_common.h"

#define RED_THREADS 256
#define NN_HV 128
#ifndef NN_WAVES      // (8 or 16 waves per 128 hand vertices: measured, no gain in the loop)
#define NN_WAVES 4
#endif
#define NN_MAX_GROUPS 64
__device__ __forceinline__ float rl_f(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// temporal smoothness (reference homan/lossutils.py:18-36): see k_smooth
__device__ __forceinline__ void smooth_body(const float* __restrict__ verts, int N, int V, int hand_nb,
                                            float* __restrict__ unit_grad, float* __restrict__ partials,
                                            unsigned int* counter, float* __restrict__ out, int out_stride, int bx, int by,
                                            int gdx)
{
    __shared__ float red[16];
    __shared__ int s_flag;
    verts += (long)by * N * (V * 3); unit_grad += (long)by * N * (V * 3);
    partials += (long)by * HM_RED_WS_FLOATS; counter += (long)by * HM_RED_WS_FLOATS;
    out += (long)by * out_stride;
    const long row = (long)V * 3, total = (long)N * row;
    const long cnt = (long)(N - hand_nb) * row;
    const float inv_cnt = cnt > 0 ? 1.0f / (float)cnt : 0.f;
    const long step = (long)hand_nb * row;
    float lsum = 0.f;
    for (long i = (long)bx * blockDim.x + threadIdx.x; i < total; i += (long)gdx * blockDim.x) {
        const int n = (int)(i / row);
        const float v = verts[i];
        float g = 0.f;
        if (n + hand_nb < N) {
            const float d = verts[i + step] - v;
            lsum += d * d;
            g -= d;
        }
        if (n - hand_nb >= 0) g += v - verts[i - step];
        unit_grad[i] = 2.0f * g * inv_cnt;
    }
    lsum = hm_block_sum(lsum, red);
    if (threadIdx.x == 0) hm_partial_store(partials + bx, lsum);
    if (hm_last_block(counter, gdx, &s_flag)) {
        const float a = hm_last_block_sum(partials, gdx, 1, red);
        if (threadIdx.x == 0) out[0] = a * inv_cnt;
    }
}

// coarse interaction loss (reference homan/losses.py:199-242): see k_inter
__device__ __forceinline__ void inter_body(const float* __restrict__ vh, const float* __restrict__ vo,
                                           const float* __restrict__ camintr, int B, int Vh, int Vo, float expansion,
                                           float zthresh, float* __restrict__ frame_rec, unsigned int* counter,
                                           float* __restrict__ out, int clip_len, int out_stride, int bx)
{
    __shared__ float red[16];
    __shared__ int s_flag;
    const int b = bx, clip = b / clip_len;
    counter += (long)clip * HM_RED_WS_FLOATS;
    const float* k = camintr + b * 9;
    // the 2 x 9 block reductions (six extrema + three sums per mesh) meet in LDS behind ONE barrier: wave results by DPP,
    // then the per-wave values combined in wave order (the order hm_block_sum uses, so the sums are the same floats)
    __shared__ float s_red[2][9][RED_THREADS / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float box[2][4], zr[2][2], cen[2][3];
    for (int which = 0; which < 2; ++which) {
        const float* v = which == 0 ? vo + (long)b * Vo * 3 : vh + (long)b * Vh * 3;
        const int V = which == 0 ? Vo : Vh;
        float umin = 3.4e38f, umax = -3.4e38f, vmin = 3.4e38f, vmax = -3.4e38f, zmin = 3.4e38f, zmax = -3.4e38f;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int i = threadIdx.x; i < V; i += blockDim.x) {
            const float x = v[3 * i], y = v[3 * i + 1], z = v[3 * i + 2];
            const float zz = z + 1e-9f;
            const float xn = x / zz, yn = (y * -1.0f) / zz;
            float u = xn * k[0] + yn * k[1];
            u = u + k[2];
            float w = xn * k[3] + yn * k[4];
            w = w + k[5];
            w = 1.0f - w;
            u = 2.0f * (u - 0.5f);
            w = 2.0f * (w - 0.5f);
            umin = fminf(umin, u); umax = fmaxf(umax, u);
            vmin = fminf(vmin, w); vmax = fmaxf(vmax, w);
            zmin = fminf(zmin, z); zmax = fmaxf(zmax, z);
            sx += x; sy += y; sz += z;
        }
        const float r9[9] = {hm_wave_min(umin), hm_wave_max(umax), hm_wave_min(vmin), hm_wave_max(vmax), hm_wave_min(zmin),
                             hm_wave_max(zmax), hm_wave_sum(sx), hm_wave_sum(sy), hm_wave_sum(sz)};
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 9; ++q) s_red[which][q][wv] = r9[q];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        for (int which = 0; which < 2; ++which) {
            const int V = which == 0 ? Vo : Vh;
            float t[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                float a = q < 6 ? s_red[which][q][0] : 0.f;
                for (int i = q < 6 ? 1 : 0; i < nw; ++i) {
                    const float x = s_red[which][q][i];
                    a = q >= 6 ? a + x : ((q & 1) ? fmaxf(a, x) : fminf(a, x));
                }
                t[q] = a;
            }
            const float cx = (t[0] + t[1]) / 2.0f, cy = (t[2] + t[3]) / 2.0f;
            const float ex = (t[1] - t[0]) / 2.0f * (1.0f + expansion), ey = (t[3] - t[2]) / 2.0f * (1.0f + expansion);
            box[which][0] = cx - ex; box[which][1] = cy - ey; box[which][2] = cx + ex; box[which][3] = cy + ey;
            zr[which][0] = t[4]; zr[which][1] = t[5];
            cen[which][0] = t[6] / (float)V; cen[which][1] = t[7] / (float)V; cen[which][2] = t[8] / (float)V;
        }
    }
    if (threadIdx.x == 0) {
        // compute_iou(box_obj, box_hand)
        const float a1 = (box[0][2] - box[0][0]) * (box[0][3] - box[0][1]);
        const float a2 = (box[1][2] - box[1][0]) * (box[1][3] - box[1][1]);
        const float w = fmaxf(fminf(box[0][2], box[1][2]) - fmaxf(box[0][0], box[1][0]), 0.f);
        const float h = fmaxf(fminf(box[0][3], box[1][3]) - fmaxf(box[0][1], box[1][1]), 0.f);
        const float inter = w * h;
        const float iou = inter / (a1 + a2 - inter);
        // compute_dist_z(verts_object, verts_hand)
        const float a = zr[0][0], bb = zr[0][1], c = zr[1][0], d = zr[1][1];
        const float zd = (d >= a && bb >= c) ? 0.f : fminf(fabsf(c - bb), fabsf(a - d));
        const float flag = ((iou > 0.f) && (zd < zthresh)) ? 1.f : 0.f;
        const float dx = cen[1][0] - cen[0][0], dy = cen[1][1] - cen[0][1], dz = cen[1][2] - cen[0][2];
        const float mse = (dx * dx + dy * dy + dz * dz) / 3.0f;
        float* r = frame_rec + b * 8;
        hm_partial_store(r, flag); hm_partial_store(r + 1, mse);
        r[2] = flag * 2.0f * dx / 3.0f; r[3] = flag * 2.0f * dy / 3.0f; r[4] = flag * 2.0f * dz / 3.0f;
    }
    if (hm_last_block(counter, clip_len, &s_flag)) {       // the clip's last frame sums the clip
        const float* fr = frame_rec + (long)clip * clip_len * 8;
        float l = 0.f;
        for (int i = threadIdx.x; i < clip_len; i += blockDim.x)
            if (hm_partial_load(fr + i * 8) != 0.f) l += hm_partial_load(fr + i * 8 + 1);
        l = hm_block_sum(l, red);
        if (threadIdx.x == 0) out[(long)clip * out_stride] = l;
    }
}

// metric-only nearest-vertex search (reference homan/losses.py:225-241): see k_nn_min
// Full nearest-vertex search (hand -> object, with indices; the body of k_nn, also a block range of the pair-terms launch).
// Virtual grid (ceil(Vh/128), B).  Workgroup = 128 hand vertices (two per lane: the pair shares every broadcast and the packed
// fp32 pipes take both) x NN_WAVES wavefronts; wave q scans an equal contiguous share of the object vertices 64 at a
// time: lane l loads object vertex l of the group (coalesced), the group is then broadcast vertex by vertex with
// v_readlane (scalar operands, no LDS round trip in the inner loop).  The partial minima are merged lexicographically
// on (distance, index) so ties keep the lowest index.
__device__ __forceinline__ void nn_full_body(const float* __restrict__ vh, const float* __restrict__ vo, int B, int Vh, int Vo,
                                             int* __restrict__ nn_idx, float* __restrict__ nn_d2, float* __restrict__ blockmin,
                                             unsigned int* counter, float* __restrict__ metric_out, int clip_len, int out_stride,
                                             int bx, int by, int gdx)
{
    __shared__ float s_d[NN_WAVES][NN_HV];
    __shared__ int s_i[NN_WAVES][NN_HV];
    __shared__ float red[16];
    __shared__ int s_flag;
    const int b = by, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    float hx[2], hy[2], hz[2], best[2];
    int besti[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = bx * NN_HV + lane + 64 * u;
        hx[u] = hy[u] = hz[u] = 0.f;
        if (i < Vh) { const float* p = vh + ((long)b * Vh + i) * 3; hx[u] = p[0]; hy[u] = p[1]; hz[u] = p[2]; }
        best[u] = 3.4e38f;
        besti[u] = 0;
    }
    const int share = (Vo + NN_WAVES - 1) / NN_WAVES, jend = min(Vo, (q + 1) * share);
    for (int j0 = q * share; j0 < jend; j0 += 64) {
        const int n = min(64, jend - j0);
        float ox = 0.f, oy = 0.f, oz = 0.f;
        if (lane < n) { const float* p = vo + ((long)b * Vo + j0 + lane) * 3; ox = p[0]; oy = p[1]; oz = p[2]; }
        int k = 0;
#define NN_STEP(K)                                                                                   \
    {                                                                                                \
        const float sx = rl_f(ox, (K)), sy = rl_f(oy, (K)), sz = rl_f(oz, (K));                      \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                              \
            const float dx = sx - hx[u], dy = sy - hy[u], dz = sz - hz[u];                           \
            const float d = dx * dx + dy * dy + dz * dz;                                             \
            const bool lt = d < best[u];                                                             \
            best[u] = lt ? d : best[u];                                                              \
            besti[u] = lt ? j0 + (K) : besti[u];                                                     \
        }                                                                                            \
    }
        for (; k + 4 <= n; k += 4) { NN_STEP(k) NN_STEP(k + 1) NN_STEP(k + 2) NN_STEP(k + 3) }
        for (; k < n; ++k) NN_STEP(k)
#undef NN_STEP
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) { s_d[q][lane + 64 * u] = best[u]; s_i[q][lane + 64 * u] = besti[u]; }
    __syncthreads();
    float bm = 3.4e38f;
    if (threadIdx.x < NN_HV) {
        const int t = threadIdx.x, i = bx * NN_HV + t;
        float bd = s_d[0][t];
        int bi = s_i[0][t];
#pragma unroll
        for (int k = 1; k < NN_WAVES; ++k) {
            const float d = s_d[k][t];
            const int id = s_i[k][t];
            if (d < bd || (d == bd && id < bi)) { bd = d; bi = id; }
        }
        if (i < Vh) { nn_idx[(long)b * Vh + i] = bi; nn_d2[(long)b * Vh + i] = bd; bm = bd; }
    }
    bm = hm_block_min(bm, red);
    // per clip (clip_len consecutive frames): its own slice of the reduce workspace, its own ticket, its own metric
    const int clip = b / clip_len, bl = b - clip * clip_len;
    blockmin += (long)clip * HM_RED_WS_FLOATS;
    counter += (long)clip * HM_RED_WS_FLOATS;
    const unsigned nblk = gdx * clip_len;
    if (threadIdx.x == 0) hm_partial_store(blockmin + bl * gdx + bx, bm);
    if (hm_last_block(counter, nblk, &s_flag)) {
        // all block minima requested at once (one agent-scope load per thread), then min over a frame's chunks, max over
        // the frames
        float* s_bm = &s_d[0][0];
        for (unsigned i2 = threadIdx.x; i2 < nblk; i2 += blockDim.x) s_bm[i2] = hm_partial_load(blockmin + i2);
        __syncthreads();
        float mx = -3.4e38f;
        for (int bb = threadIdx.x; bb < clip_len; bb += blockDim.x) {
            float m = 3.4e38f;
            for (int c = 0; c < gdx; ++c) m = fminf(m, s_bm[bb * gdx + c]);
            mx = fmaxf(mx, sqrtf(m));
        }
        mx = hm_block_max(mx, red);
        if (threadIdx.x == 0) metric_out[(long)clip * out_stride] = mx;
    }
}


#define NN_SEED_FRAMES 128       // frames of a clip the seed bookkeeping of nn_min_body covers (longer clips search unseeded)
#ifdef NN_PHASES
static __device__ unsigned long long g_nn_ph[10];     // wall-clock ticks of thread 0, summed over workgroups: loads+spheres, bounds, list, scan, reduce+ticket, finish; workgroups; survivors
#define NNP_MARK(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_nn_ph[k], t_ - nnp_t); nnp_t = t_; } } while (0)
#else
#define NNP_MARK(k)
#endif
__device__ __forceinline__ void nn_min_body(const float* __restrict__ vh, const float* __restrict__ vo, int B, int Vh, int Vo,
                                            float* __restrict__ blockmin, unsigned int* counter,
                                            float* __restrict__ metric_out, int clip_len, int out_stride,
                                            const int* __restrict__ obj_order, int bx, int by, int gdx,
                                            const float* __restrict__ sph_mesh = nullptr,
                                            const float* __restrict__ obj_rot6d = nullptr,
                                            const float* __restrict__ obj_trans = nullptr,
                                            const float* __restrict__ obj_scale = nullptr,
                                            const int* __restrict__ hand_order = nullptr,
                                            int* __restrict__ seed = nullptr)
{
    // `seed` (optional, 2 B + gdx B ints, zero-filled once by the caller, carried from launch to launch): per frame the vertex
    // pair (hand i, object j) that held the frame's minimum at the LAST launch.  Its distance at the CURRENT positions is the
    // distance of a real pair, i.e. an upper bound of this launch's minimum, known before anything is scanned: the workgroups
    // then scan only the groups that can beat it - for most chunks of the hand none - instead of finding a bound of their own
    // in four exact scans each.  Scheduling data only: any pair gives a valid bound, the result stays the exact minimum.
    __shared__ float s_sph[NN_MAX_GROUPS][4];
    __shared__ float s_lb[NN_MAX_GROUPS];
    __shared__ unsigned s_ub;
    __shared__ int s_first[NN_WAVES];
    __shared__ int s_list[NN_MAX_GROUPS], s_n;
    __shared__ float s_d[NN_WAVES][NN_HV];
    __shared__ float red[16];
    __shared__ int s_flag;
    const int b = by, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int ng = (Vo + 63) >> 6;
#ifdef NN_PHASES
    unsigned long long nnp_t = wall_clock64();
#endif
    float hx[2], hy[2], hz[2];
    bool hv[2];
    int hraw[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = bx * NN_HV + lane + 64 * u;
        hv[u] = i < Vh;
        hx[u] = hy[u] = hz[u] = 0.f;
        // (hand_order: a spatial sort of the hand's vertices - the minimum does not care which vertex sits in which lane, and
        //  a workgroup whose 128 vertices are one patch of the hand instead of a sample of all of it has few groups in reach)
        if (hv[u]) {
            hraw[u] = hand_order ? hand_order[i] : i;
            const float* p = vh + ((long)b * Vh + hraw[u]) * 3; hx[u] = p[0]; hy[u] = p[1]; hz[u] = p[2];
        }
    }
    bool seeded = false;
    unsigned seed_bits = 0x7f7fffffu;
    if (seed) {
        const int si = seed[2 * b], sj = seed[2 * b + 1];          // (uniform)
        if (si >= 0 && si < Vh && sj >= 0 && sj < Vo) {
            const float* ph = vh + ((long)b * Vh + si) * 3;
            const float* po = vo + ((long)b * Vo + sj) * 3;
            const float dx = po[0] - ph[0], dy = po[1] - ph[1], dz = po[2] - ph[2];      // (the scans' operand order)
            seed_bits = __float_as_uint(dx * dx + dy * dy + dz * dz);
            seeded = true;
        }
    }
    if (threadIdx.x == 0) { s_ub = seed_bits; s_n = 0; }      // (s_ub: smallest exact squared distance known before the list is built)
    if (threadIdx.x < NN_WAVES) s_first[threadIdx.x] = -1;
    if (sph_mesh && (int)threadIdx.x < ng) {
        // the object is rigid: the bounding spheres of its groups are a table in MESH space (built once by the caller), and a
        // lane per group carries centre and radius into this frame's camera space - instead of every workgroup loading and
        // reducing all the groups' vertices again (in an 8-clip batch that was 15 M wave-instructions per launch, next to the
        // line expansion it shares the GPU with)
        float R[9];
        rot6d_to_mat(obj_rot6d + b * 6, R);
        const float sc = fabsf(obj_scale[b / clip_len]);
        const float* c = sph_mesh + ((long)b * ng + threadIdx.x) * 4;
        const float x = sc * c[0], y = sc * c[1], z = sc * c[2];
        const float* t = obj_trans + b * 3;
        s_sph[threadIdx.x][0] = x * R[0] + y * R[3] + z * R[6] + t[0];
        s_sph[threadIdx.x][1] = x * R[1] + y * R[4] + z * R[7] + t[1];
        s_sph[threadIdx.x][2] = x * R[2] + y * R[5] + z * R[8] + t[2];
        s_sph[threadIdx.x][3] = sc * c[3] * (1.0f + 1e-4f) + 1e-6f;      // (slack for the rounding of both transforms)
    }
    __syncthreads();
    NNP_MARK(0);
    // 1 + 2: spheres and bounds of this wave's groups
    for (int g = q; g < ng; g += NN_WAVES) {
        float cx, cy, cz, rg;
        if (sph_mesh) { cx = s_sph[g][0]; cy = s_sph[g][1]; cz = s_sph[g][2]; rg = s_sph[g][3]; }
        else {
            const int j = 64 * g + lane, n = min(64, Vo - 64 * g);
            float ox = 0.f, oy = 0.f, oz = 0.f;
            if (lane < n) { const float* p = vo + ((long)b * Vo + (obj_order ? obj_order[j] : j)) * 3; ox = p[0]; oy = p[1]; oz = p[2]; }
            const float inv_n = 1.0f / (float)n;
            cx = hm_wave_sum(ox) * inv_n; cy = hm_wave_sum(oy) * inv_n; cz = hm_wave_sum(oz) * inv_n;
            const float ex = ox - cx, ey = oy - cy, ez = oz - cz;
            rg = sqrtf(hm_wave_max(lane < n ? ex * ex + ey * ey + ez * ez : 0.f)) * (1.0f + 1e-5f);
        }
        float dc = 3.4e38f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (hv[u]) {
                const float dx = cx - hx[u], dy = cy - hy[u], dz = cz - hz[u];
                dc = fminf(dc, sqrtf(dx * dx + dy * dy + dz * dz));
            }
        dc = hm_wave_min(dc);
        if (lane == 0) s_lb[g] = dc * (1.0f - 1e-5f) - rg;      // no vertex of the group is closer to any of the 128
    }
    __syncthreads();
    NNP_MARK(1);
    // 3: the NN_WAVES groups with the smallest lower bounds are scanned first, one per wave: the smallest exact distance
    // found there is a far tighter upper bound than "centre distance + radius" (with a hand that touches the object nearly
    // every group passed that test: 21.5 of 24 at cfg2) ...
    int rank = NN_MAX_GROUPS;
    if (!seeded && (int)threadIdx.x < ng) {
        const float mine = s_lb[threadIdx.x];
        rank = 0;
        for (int g = 0; g < ng; ++g) {
            const float o = s_lb[g];
            rank += (o < mine || (o == mine && g < (int)threadIdx.x)) ? 1 : 0;
        }
    }
    __syncthreads();
    if (rank < NN_WAVES) { s_first[rank] = threadIdx.x; s_lb[threadIdx.x] = 3.4e38f; }      // (scanned below: never listed again)
    __syncthreads();
    // running minima as the BITS of the squared distances (non-negative floats order like their bits: one integer min per
    // pair instead of a NaN-quieting float min), four object vertices per trip with the trip count in a scalar: a wave is
    // alone on its SIMD here, so the four independent chains are what hides the arithmetic latency
    unsigned bestu[2] = {0x7f7fffffu, 0x7f7fffffu};
    int bestg[2] = {0, 0};                  // the group a lane's current minimum came from (for the next launch's seed)
    auto scan_group = [&](const int g) {
        const unsigned before0 = bestu[0], before1 = bestu[1];
        const int j = 64 * g + lane, n = __builtin_amdgcn_readfirstlane(min(64, Vo - 64 * g));
        float ox = 0.f, oy = 0.f, oz = 0.f;
        if (lane < n) { const float* p = vo + ((long)b * Vo + (obj_order ? obj_order[j] : j)) * 3; ox = p[0]; oy = p[1]; oz = p[2]; }
        auto step = [&](const int k) {
            const float sx = rl_f(ox, k), sy = rl_f(oy, k), sz = rl_f(oz, k);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float dx = sx - hx[u], dy = sy - hy[u], dz = sz - hz[u];
                bestu[u] = min(bestu[u], __float_as_uint(dx * dx + dy * dy + dz * dz));
            }
        };
        int k = 0;
        for (; k + 4 <= n; k += 4) { step(k); step(k + 1); step(k + 2); step(k + 3); }
        for (; k < n; ++k) step(k);
        bestg[0] = bestu[0] < before0 ? g : bestg[0];
        bestg[1] = bestu[1] < before1 ? g : bestg[1];
    };
    if (s_first[q] >= 0) {
        scan_group(s_first[q]);
        const float wm = hm_wave_min(fminf(hv[0] ? __uint_as_float(bestu[0]) : 3.4e38f, hv[1] ? __uint_as_float(bestu[1]) : 3.4e38f));
        if (lane == 0) atomicMin(&s_ub, __float_as_uint(wm));          // squared distance, positive
    }
    __syncthreads();
    // ... 4: and only the groups whose lower bound does not exceed it are scanned as well
    const float ub_exact = sqrtf(__uint_as_float(s_ub)) * (1.0f + 1e-5f) + 1e-7f;
    if ((int)threadIdx.x < ng && s_lb[threadIdx.x] <= ub_exact) s_list[atomicAdd(&s_n, 1)] = threadIdx.x;
    __syncthreads();
    const int ns = s_n;
    NNP_MARK(2);
#ifdef NN_PHASES
    if (threadIdx.x == 0) { atomicAdd(&g_nn_ph[6], 1ull); atomicAdd(&g_nn_ph[7], (unsigned long long)ns); }
#endif
    for (int e = q; e < ns; e += NN_WAVES) scan_group(s_list[e]);
    float bm = fminf(hv[0] ? __uint_as_float(bestu[0]) : 3.4e38f, hv[1] ? __uint_as_float(bestu[1]) : 3.4e38f);
#ifdef NN_PHASES
    __syncthreads();
    NNP_MARK(3);
#endif
    bm = hm_block_min(bm, red);
    const int clip = b / clip_len, bl = b - clip * clip_len;
    blockmin += (long)clip * HM_RED_WS_FLOATS;
    counter += (long)clip * HM_RED_WS_FLOATS;
    const unsigned nblk = gdx * clip_len;
    __shared__ unsigned s_pick;
    const bool seeding = seed && clip_len <= NN_SEED_FRAMES;
    if (seeding) {
        // which of this workgroup's vertices holds its minimum, and which group it came from: (raw hand vertex << 6 | group),
        // any holder in a tie - the clip's last workgroup turns the frame's winner into the next launch's seed pair
        if (threadIdx.x == 0) s_pick = 0xffffffffu;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (hv[u] && bm < 3.0e38f && __uint_as_float(bestu[u]) == bm) atomicMin(&s_pick, ((unsigned)hraw[u] << 6) | (unsigned)bestg[u]);
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(seed + 2L * B + (long)b * gdx + bx, (int)s_pick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) hm_partial_store(blockmin + bl * gdx + bx, bm);
    const bool nn_last_blk = hm_last_block(counter, nblk, &s_flag);
    NNP_MARK(4);
    if (nn_last_blk) {
        float* s_bm = &s_d[0][0];
        __shared__ int s_rec[NN_SEED_FRAMES];
        __shared__ unsigned s_mb[NN_SEED_FRAMES];
        for (unsigned i2 = threadIdx.x; i2 < nblk; i2 += blockDim.x) s_bm[i2] = hm_partial_load(blockmin + i2);
        __syncthreads();
        float mx = -3.4e38f;
        for (int bb = threadIdx.x; bb < clip_len; bb += blockDim.x) {
            float m = 3.4e38f;
            unsigned cbest = 0;
            for (unsigned c = 0; c < gdx; ++c) {
                const float v = s_bm[bb * gdx + c];
                if (v < m) { m = v; cbest = c; }          // (the same minimum as a chain of fminf: no NaN among squared distances)
            }
            mx = fmaxf(mx, sqrtf(m));
            if (seeding) {
                s_mb[bb] = __float_as_uint(m);
                s_rec[bb] = m < 3.0e38f ? __hip_atomic_load(seed + 2L * B + ((long)clip * clip_len + bb) * gdx + cbest, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT) : -1;
            }
        }
        mx = hm_block_max(mx, red);
        if (threadIdx.x == 0) metric_out[(long)clip * out_stride] = mx;
        if (seeding) {
            // the winner's object vertex: the one of its group at exactly the winning distance (a thread per (frame, vertex of the group))
            __syncthreads();
            for (int t = threadIdx.x; t < clip_len * 64; t += blockDim.x) {
                const int bb = t >> 6, k = t & 63, rec = s_rec[bb];
                const long fb = (long)clip * clip_len + bb;
                const int j = 64 * (rec & 63) + k;
                if (rec < 0 || j >= Vo) continue;
                const int jj = obj_order ? obj_order[j] : j, ii = rec >> 6;
                const float* po = vo + (fb * Vo + jj) * 3;
                const float* ph = vh + (fb * Vh + ii) * 3;
                const float dx = po[0] - ph[0], dy = po[1] - ph[1], dz = po[2] - ph[2];
                if (__float_as_uint(dx * dx + dy * dy + dz * dz) == s_mb[bb]) { seed[2 * fb] = ii; seed[2 * fb + 1] = jj; }
            }
        }
    }
}

// 2-D reprojection + temporal smoothness of the hand vertices + priors: see k_hand_terms
__device__ __forceinline__ void hand_terms_body(
    const float* __restrict__ verts, const float* __restrict__ camintr, int hand_nb, const float* __restrict__ ref2d,
    float image_size, int N, int V, float* __restrict__ unit_v2d, float* __restrict__ out_v2d,
    float* __restrict__ unit_smooth, float* __restrict__ out_smooth, const float* __restrict__ pca, long npca,
    const float* __restrict__ s_obj, const float* __restrict__ m_obj, const float* __restrict__ s_hand,
    const float* __restrict__ m_hand, float* __restrict__ g_pca, float* __restrict__ g_sobj,
    float* __restrict__ g_shand, float* __restrict__ out_priors, float* __restrict__ partials, unsigned int* counter,
    int out_stride, int bx, int by, int gdx)
{
    __shared__ float red[16];
    __shared__ int s_flag;
    verts += (long)by * N * (V * 3); ref2d += (long)by * N * (V * 2); unit_v2d += (long)by * N * (V * 3); unit_smooth += (long)by * N * (V * 3);
    camintr += (long)by * (N / hand_nb) * 9;
    partials += (long)by * HM_RED_WS_FLOATS; counter += (long)by * HM_RED_WS_FLOATS;
    out_v2d += (long)by * out_stride;
    out_smooth += (long)by * out_stride;
    if (pca) {
        pca += (long)by * npca; g_pca += (long)by * npca;
        s_obj += by; m_obj += by; s_hand += by; m_hand += by;
        g_sobj += by; g_shand += by;
        out_priors += (long)by * out_stride;
    }
    // ---- v2d
    const long total = (long)N * V;
    const float inv_cnt = 1.0f / (float)total;
    float lsum = 0.f, msum = 0.f;
    for (long i = (long)bx * blockDim.x + threadIdx.x; i < total; i += (long)gdx * blockDim.x) {
        const int n = (int)(i / V);
        const float* k = camintr + (n / hand_nb) * 9;
        const float x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
        const float hx = k[0] * x + k[1] * y + k[2] * z;
        const float hy = k[3] * x + k[4] * y + k[5] * z;
        const float hz = k[6] * x + k[7] * y + k[8] * z;
        const float px = hx / hz, py = hy / hz;
        const float rx = ref2d[2 * i], ry = ref2d[2 * i + 1];
        const float dx = px - rx / image_size, dy = py - ry / image_size;
        lsum += dx * dx + dy * dy;
        const float mx = px * image_size - rx, my = py * image_size - ry;
        msum += sqrtf(mx * mx + my * my);
        const float gpx = 2.0f * dx * inv_cnt, gpy = 2.0f * dy * inv_cnt;
        const float ghx = gpx / hz, ghy = gpy / hz, ghz = -(gpx * hx + gpy * hy) / (hz * hz);
        unit_v2d[3 * i] = k[0] * ghx + k[3] * ghy + k[6] * ghz;
        unit_v2d[3 * i + 1] = k[1] * ghx + k[4] * ghy + k[7] * ghz;
        unit_v2d[3 * i + 2] = k[2] * ghx + k[5] * ghy + k[8] * ghz;
    }
    // ---- temporal smoothness
    const long row = (long)V * 3, etotal = (long)N * row;
    const long cnt = (long)(N - hand_nb) * row;
    const float sinv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
    const long step = (long)hand_nb * row;
    float ssum = 0.f;
    for (long i = (long)bx * blockDim.x + threadIdx.x; i < etotal; i += (long)gdx * blockDim.x) {
        const int n = (int)(i / row);
        const float v = verts[i];
        float g = 0.f;
        if (n + hand_nb < N) {
            const float d = verts[i + step] - v;
            ssum += d * d;
            g -= d;
        }
        if (n - hand_nb >= 0) g += v - verts[i - step];
        unit_smooth[i] = 2.0f * g * sinv;
    }
    lsum = hm_block_sum(lsum, red);
    msum = hm_block_sum(msum, red);
    ssum = hm_block_sum(ssum, red);
    // ---- priors (block 0)
    if (pca && bx == 0) {
        float a = 0.f;
        const float inv = 1.0f / (float)npca;
        for (long i = threadIdx.x; i < npca; i += blockDim.x) {
            const float p = pca[i];
            a += p * p;
            g_pca[i] = 2.0f * p * inv;
        }
        a = hm_block_sum(a, red);
        if (threadIdx.x == 0) {
            out_priors[0] = a * inv;
            const float d0 = s_obj[0] - m_obj[0], d1 = s_hand[0] - m_hand[0];
            out_priors[1] = d0 * d0;
            out_priors[2] = d1 * d1;
            g_sobj[0] = 2.0f * d0;
            g_shand[0] = 2.0f * d1;
        }
    }
    if (threadIdx.x == 0) {
        hm_partial_store(partials + 3 * bx, lsum);
        hm_partial_store(partials + 3 * bx + 1, msum);
        hm_partial_store(partials + 3 * bx + 2, ssum);
    }
    if (hm_last_block(counter, gdx, &s_flag)) {
        const float a = hm_last_block_sum(partials, gdx, 3, red);
        const float b = hm_last_block_sum(partials + 1, gdx, 3, red);
        const float c = hm_last_block_sum(partials + 2, gdx, 3, red);
        if (threadIdx.x == 0) { out_v2d[0] = a * inv_cnt; out_v2d[1] = b * inv_cnt; out_smooth[0] = c * sinv; }
    }
}
