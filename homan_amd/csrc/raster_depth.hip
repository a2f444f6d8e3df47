// raster_depth.hip -- depth-image backward (NMR backward_depth_map) and the ordinal depth term (PHOSA)
#include "raster_ws.h"

// ---------------------------------------------------------------- depth-image backward (NMR backward_depth_map)
// d pooled_depth / d face vertices, analytic: per covered sample of a face, with zp its depth and w_k its (clamped,
// renormalised) barycentrics,  dz_k += g w_k zp^2 / z_k^2  and  d(x,y)_k += -g w_k zp^2 tmp[l] is/2  with
// tmp[l] = -sum_m inv[m][l] / z_m.  Both factor through A_k = sum_samples g zp^2 w_k, so a wave per (frame, face)
// strides the face's sample box, tests ownership in the index map, and reduces three numbers per winding.
// gf9 (B,F,2,9): gradient w.r.t. the NDC vertices in WINDING order.
#ifndef DBF_FACES
#define DBF_FACES 4        // consecutive (frame, face) slots per wave
#endif
__global__ __launch_bounds__(256) void k_depth_bwd_faces(const float* __restrict__ faces9, const FaceBox* __restrict__ boxes,
                                                         const int* __restrict__ idx_map, const float* __restrict__ gpd,
                                                         const unsigned char* __restrict__ owned, int B, int F, int S,
                                                         float* __restrict__ gf9, const unsigned char* __restrict__ gflags)
{
    // A wave takes DBF_FACES consecutive face slots.  Half the windings own no sample (hidden, back-facing, culled): one lane
    // per (slot, winding) reads box mask and ownership flag - one coalesced round trip for the run - and writes the nine
    // zeros of an idle winding itself; the wave then walks only the windings that own something, each exactly as the
    // one-wave-per-face launch did (same lanes, same sums).  That launch was 90 000 waves for the bottle, most of them a
    // dependent chain of three round trips to find out they had nothing to do.
    const int lane = threadIdx.x & 63;
    const int is = 2 * S;
    const long nbf = (long)B * F;
    const long w0 = (long)__builtin_amdgcn_readfirstlane((int)(((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6)) * DBF_FACES;
    if (w0 >= nbf) return;
    bool live = false;
    if (lane < 2 * DBF_FACES) {
        const long bfl = w0 + (lane >> 1);
        const int var = lane & 1;
        if (bfl < nbf) {
            const int bl = (int)(bfl / F), fl = (int)(bfl % F);
            const unsigned m = (reinterpret_cast<const uint2*>(boxes)[bfl].x >> 14) & 3u;
            live = ((m >> var) & 1u) && owned[(long)bl * 2 * F + fl + var * F];
            if (live && gflags) {
                // SPARSE upstream gradient (an ordinal depth term is zero wherever render and annotation agree on the order:
                // nearly everywhere): gflags holds one byte per (frame, pixel row, 64-pixel segment) = "some pixel of the segment
                // has a non-zero gradient" (hm_ordinal_depth_bwd_flags).  A winding whose sample box touches no flagged segment
                // would sum exact zeros over its samples: it gets its zeros here, without the walk.  (Boxes taller than 64
                // pixel rows are walked regardless.)
                const uint2 fb = reinterpret_cast<const uint2*>(boxes)[bfl];
                const int bx0 = fb.x & 0x3fff, by0 = (int)(fb.x >> 16), bx1 = (int)(fb.y & 0xffff), by1 = (int)(fb.y >> 16);
                const int r0 = (is - 1 - by1) >> 1, r1 = (is - 1 - by0) >> 1, s0 = bx0 >> 7, s1 = bx1 >> 7, nseg = S >> 6;
                if (r1 - r0 < 64) {
                    unsigned any = 0u;
                    for (int r = r0; r <= r1; ++r)
                        for (int sg = s0; sg <= s1; ++sg) any |= gflags[((long)bl * S + r) * nseg + sg];
                    live = any != 0u;
                }
            }
            if (!live) {
                float* out = gf9 + (bfl * 2 + var) * 9;
#pragma unroll
                for (int k = 0; k < 9; ++k) out[k] = 0.f;
            }
        }
    }
    unsigned long long todo = __ballot(live);
    while (todo) {
        const int t = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long bf = w0 + (t >> 1);
        const int var = t & 1;
        const int b = (int)(bf / F), fi = (int)(bf % F);
        const uint2 bx = reinterpret_cast<const uint2*>(boxes)[bf];
        const int x0 = bx.x & 0x3fff, y0 = (int)(bx.x >> 16), x1 = (int)(bx.y & 0xffff), y1 = (int)(bx.y >> 16);
        const float* src = faces9 + bf * 9;
        const int* idx = idx_map + (long)b * is * is;
        const float* g = gpd + (long)b * S * S;
        float* out = gf9 + (bf * 2 + var) * 9;
        const int fn = fi + var * F;
        float f[9];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int sv = var ? 2 - k : k;
            f[3 * k] = src[3 * sv]; f[3 * k + 1] = src[3 * sv + 1]; f[3 * k + 2] = src[3 * sv + 2];
        }
        float p[3][2];
#pragma unroll
        for (int k = 0; k < 3; ++k) { p[k][0] = topix(f[3 * k], is); p[k][1] = topix(f[3 * k + 1], is); }
        float inv[9] = {
            p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
            p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
            p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
        const float den = p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) + p[1][0] * (p[2][1] - p[0][1]);
#pragma unroll
        for (int k = 0; k < 9; ++k) inv[k] = inv[k] / den;
        const float rz0 = 1.0f / f[2], rz1 = 1.0f / f[5], rz2 = 1.0f / f[8];
        const int bw = x1 - x0 + 1, n = bw * (y1 - y0 + 1);
        float A0 = 0.f, A1 = 0.f, A2 = 0.f;
        for (int e = lane; e < n; e += 64) {
            const int xi = x0 + e % bw, yi = y0 + e / bw;
            // (the upstream gradient of the sample is requested WITH its owner, not behind the test)
            const int owner = idx[(long)yi * is + xi];
            const float gs = g[(long)((is - 1 - yi) >> 1) * S + (xi >> 1)];
            if (owner != fn) continue;
            float wgt[3], ws = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float tt = inv[3 * k] * (float)xi;
                tt = tt + inv[3 * k + 1] * (float)yi;
                tt = tt + inv[3 * k + 2];
                tt = fminf(fmaxf(tt, 0.0f), 1.0f);
                wgt[k] = tt;
                ws += tt;
            }
            float sum = wgt[0] * rz0;
            sum = sum + wgt[1] * rz1;
            sum = sum + wgt[2] * rz2;
            const float zp = ws / sum;
            const float a = 0.25f * gs * zp * zp;
            A0 += a * (wgt[0] / ws); A1 += a * (wgt[1] / ws); A2 += a * (wgt[2] / ws);
        }
        A0 = hm_wave_sum(A0); A1 = hm_wave_sum(A1); A2 = hm_wave_sum(A2);
        if (lane == 0) {
            const float tmp0 = -(inv[0] * rz0 + inv[3] * rz1 + inv[6] * rz2);
            const float tmp1 = -(inv[1] * rz0 + inv[4] * rz1 + inv[7] * rz2);
            const float A[3] = {A0, A1, A2}, rz[3] = {rz0, rz1, rz2};
            const float half_is = 0.5f * (float)is;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                out[3 * k] = -A[k] * tmp0 * half_is;
                out[3 * k + 1] = -A[k] * tmp1 * half_is;
                out[3 * k + 2] = A[k] * rz[k] * rz[k];
            }
        }
    }
}

// vertex gather of gf9 (winding order -> mesh corners) + projection backward (z passes straight through)
__global__ void k_depth_bwd_gather(const float* __restrict__ gf9, const int* __restrict__ adj_off,
                                   const int* __restrict__ adj_items, const float* __restrict__ verts,
                                   const float* __restrict__ K, int B, int V, int F, float orig_size,
                                   float* __restrict__ grad_verts, const unsigned char* __restrict__ gflags, int S)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gflags) {
        // a frame without any flagged segment (see k_depth_bwd_faces) has gf9 == 0 throughout: its vertices get their zeros
        // without the two dependent round trips of the gather.  The workgroup scans the flag bytes of the (one or two) frames
        // its vertices belong to.
        __shared__ int s_any[8];
        const long nv = (long)B * V;
        const int b_lo = (int)(((long)blockIdx.x * blockDim.x) / V);
        const int b_hi = (int)(min((long)blockIdx.x * blockDim.x + blockDim.x - 1, nv - 1) / V);
        if (threadIdx.x < 8) s_any[threadIdx.x] = b_hi - b_lo >= 8 ? 1 : 0;
        __syncthreads();
        if (b_hi - b_lo < 8) {
            const int nflag = S * (S >> 6);                               // bytes per frame, a multiple of 64
            for (int bb = b_lo; bb <= b_hi; ++bb) {
                unsigned a = 0u;
                for (int k = 4 * (int)threadIdx.x; k < nflag; k += 4 * (int)blockDim.x)
                    a |= *reinterpret_cast<const unsigned*>(gflags + (long)bb * nflag + k);
                if (a) s_any[bb - b_lo] = 1;
            }
        }
        __syncthreads();
        if (i >= nv) return;
        if (!s_any[(int)(i / V) - b_lo]) {
            grad_verts[3 * i] = 0.f; grad_verts[3 * i + 1] = 0.f; grad_verts[3 * i + 2] = 0.f;
            return;
        }
    }
    if (i >= (long)B * V) return;
    const int b = (int)(i / V), v = (int)(i % V);
    float gu = 0.f, gv = 0.f, gz = 0.f;
    // a vertex's corners eight at a time: their item numbers in one round trip, their 6 x 8 gradient words in the next, then
    // the additions in the adjacency's order (the same sums as the corner-by-corner walk, whose 2 x valence dependent round
    // trips made this 45 000-thread launch 24 us long)
    const int a_beg = adj_off[v], a_end = adj_off[v + 1];
    const float* kk = K + b * 9;
    const float x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
    const float k0 = kk[0], k1 = kk[1], k3 = kk[3], k4 = kk[4];
    for (int a = a_beg; a < a_end; a += 8) {
        int item[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) item[j] = adj_items[min(a + j, a_end - 1)];
        float val[8][6];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int fi = item[j] / 3, k = item[j] % 3;
            const float* pf = gf9 + ((long)b * F + fi) * 18;
            val[j][0] = pf[3 * k]; val[j][1] = pf[3 * k + 1]; val[j][2] = pf[3 * k + 2];
            val[j][3] = pf[9 + 3 * (2 - k)]; val[j][4] = pf[9 + 3 * (2 - k) + 1]; val[j][5] = pf[9 + 3 * (2 - k) + 2];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (a + j < a_end) {
                gu += val[j][0] + val[j][3];
                gv += val[j][1] + val[j][4];
                gz += val[j][2] + val[j][5];
            }
    }
    const float zz = z + 1e-9f;
    const float du0 = gu * (2.0f / orig_size), dv0 = -gv * (2.0f / orig_size);
    const float dxn = k0 * du0 + k3 * dv0;
    const float dyn = k1 * du0 + k4 * dv0;
    grad_verts[3 * i] = dxn / zz;
    grad_verts[3 * i + 1] = dyn / zz;
    grad_verts[3 * i + 2] = -(dxn * x + dyn * y) / (zz * zz) + gz;
}

// ---------------------------------------------------------------- ordinal depth loss (PHOSA), two layers
// reference homan/lossutils.py:133-169 as the method intends (the reference code itself cannot run: see DESIGN.md).
// layers 0 = object, 1 = hand; d*/a* = pooled depth / alpha renders (B,S,S); m* = instance masks (B,S,S) uint8.
// rec (5 floats): num_pairs, msum01, S01, msum10, S10.
// grid (ORD_CHUNKS, B): a frame's pixels are split over ORD_CHUNKS workgroups (one workgroup per frame walked 256 pixels per
// thread behind four to six dependent loads each: 95 us for 30 frames of 256^2, the longest launch of the depth term).  The
// frame record (8 words of frame_part, ZERO on entry, re-zeroed by the finishing workgroup) collects the chunks with 64-bit
// integer atomics - exact and order-independent, so the result does not depend on which chunk lands first:
//   words 0-1  pixels of layer 0 | layer 1 << 21 | both << 42   (21 bits each: S <= 1024)
//   words 2-3  pixels ordered wrongly: (annotated 0 in front) | (annotated 1 in front) << 21 | chunks arrived << 42
//   words 4-5 / 6-7  softplus sums of the two kinds, fixed point 2^-32 (a workgroup's own float sum, then integer adds)
#define ORD_CHUNKS 16
#define ORD_FIX 4294967296.0       // 2^32
#define ORD_MAXB 256               // frames whose records the finishing workgroup stages in LDS (longer clips: thread 0 alone)
__global__ __launch_bounds__(256) void k_ordinal_depth(const float* __restrict__ d0, const float* __restrict__ d1,
                                                        const float* __restrict__ a0, const float* __restrict__ a1,
                                                        const unsigned char* __restrict__ m0,
                                                        const unsigned char* __restrict__ m1, int B, int S,
                                                        float* __restrict__ frame_part, unsigned int* counter,
                                                        float* __restrict__ rec, float* __restrict__ out)
{
    __shared__ float red[16 * 7];
    __shared__ int s_flag;
    const int b = blockIdx.y;
    const long base = (long)b * S * S;
    const int per = (S * S + ORD_CHUNKS - 1) / ORD_CHUNKS, i0 = blockIdx.x * per, i1 = min(S * S, i0 + per);
    float c00 = 0.f, c11 = 0.f, c01 = 0.f, ms01 = 0.f, s01 = 0.f, ms10 = 0.f, s10 = 0.f;
    // four of a thread's pixels per trip, all six words of each requested before the first is looked at: 4 dependent round
    // trips per thread where the pixel-by-pixel walk (alpha -> test -> depths and masks -> test) had 32.  Same pixels in the
    // same order per thread: the same sums.
    for (int i = i0 + threadIdx.x; i < i1; i += 4 * blockDim.x) {
        float av0[4], av1[4], zv0[4], zv1[4];
        unsigned char mv0[4], mv1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long at = base + min(i + u * (int)blockDim.x, i1 - 1);
            av0[u] = a0[at]; av1[u] = a1[at]; zv0[u] = d0[at]; zv1[u] = d1[at]; mv0[u] = m0[at]; mv1[u] = m1[at];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + u * (int)blockDim.x >= i1) break;
            const bool s0 = av0[u] == 1.0f, s1 = av1[u] == 1.0f;
            c00 += s0 ? 1.f : 0.f; c11 += s1 ? 1.f : 0.f;
            if (s0 && s1) {
                c01 += 1.f;
                const float z0 = zv0[u], z1 = zv1[u];
                const bool g0 = mv0[u] != 0, g1 = mv1[u] != 0;
                if (g0 && !g1 && z1 < z0) { ms01 += 1.f; s01 += logf(1.0f + expf(fminf(fmaxf(z0 - z1, 0.f), 2.f))); }
                if (g1 && !g0 && z0 < z1) { ms10 += 1.f; s10 += logf(1.0f + expf(fminf(fmaxf(z1 - z0, 0.f), 2.f))); }
            }
        }
    }
    float v[7] = {c00, c11, c01, ms01, s01, ms10, s10};
    hm_block_sum_n<7>(v, red);           // (two barriers for the seven sums; the combination order of hm_block_sum)
    unsigned long long* fr = reinterpret_cast<unsigned long long*>(frame_part) + (long)b * 4;
    // the frame's record; its word 1 also counts the chunks that have arrived (bits 42..), so the chunk that completes a frame
    // learns it from the atomic it issues anyway, and only ONE workgroup per frame draws the clip's ticket (a ticket per chunk
    // was ORD_CHUNKS * B returning atomics on one word: they serialise)
    if (threadIdx.x == 0) {
        atomicAdd(fr, (unsigned long long)v[0] | ((unsigned long long)v[1] << 21) | ((unsigned long long)v[2] << 42));
        atomicAdd(fr + 2, (unsigned long long)((double)v[4] * ORD_FIX));
        atomicAdd(fr + 3, (unsigned long long)((double)v[6] * ORD_FIX));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long old = atomicAdd(fr + 1, (unsigned long long)v[3] | ((unsigned long long)v[5] << 21) | (1ull << 42));
        s_flag = (int)(old >> 42) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_flag) return;               // (block-uniform)
    if (hm_last_block(counter, gridDim.y, &s_flag)) {
        // the frames in frame order (fixed), one thread: B is a clip's length.  The frames' records are fetched (and re-armed) by
        // a thread each first - thread 0 walking them one agent-scope load after the other was B dependent round trips, most of
        // this launch's time
        __shared__ unsigned long long s_w[4][ORD_MAXB];
        unsigned long long* all = reinterpret_cast<unsigned long long*>(frame_part);
        const bool staged = B <= ORD_MAXB;
        if (staged) {
            for (int f = threadIdx.x; f < B; f += blockDim.x) {
#pragma unroll
                for (int k = 0; k < 4; ++k) s_w[k][f] = __hip_atomic_load(all + 4L * f + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                all[4L * f] = 0ull; all[4L * f + 1] = 0ull; all[4L * f + 2] = 0ull; all[4L * f + 3] = 0ull;     // re-armed
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            float t[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            for (int f = 0; f < B; ++f) {
                const unsigned long long w0 = staged ? s_w[0][f] : __hip_atomic_load(all + 4L * f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long w1 = staged ? s_w[1][f] : __hip_atomic_load(all + 4L * f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long w2 = staged ? s_w[2][f] : __hip_atomic_load(all + 4L * f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long w3 = staged ? s_w[3][f] : __hip_atomic_load(all + 4L * f + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool h0 = (w0 & 0x1fffffull) != 0ull, h1 = ((w0 >> 21) & 0x1fffffull) != 0ull, h01 = (w0 >> 42) != 0ull;
                t[0] += (h0 ? 1.f : 0.f) + (h1 ? 1.f : 0.f) + 2.f * (h01 ? 1.f : 0.f);        // pairs of this frame
                t[1] += (float)(unsigned)(w1 & 0x1fffffull);
                t[2] += (float)((double)w2 / ORD_FIX);
                t[3] += (float)(unsigned)((w1 >> 21) & 0x1fffffull);
                t[4] += (float)((double)w3 / ORD_FIX);
                if (!staged) { all[4L * f] = 0ull; all[4L * f + 1] = 0ull; all[4L * f + 2] = 0ull; all[4L * f + 3] = 0ull; }     // re-armed
            }
            float loss = 0.f;
            if (t[1] > 0.f) loss += t[2] / t[1];
            if (t[3] > 0.f) loss += t[4] / t[3];
            out[0] = loss / t[0];
#pragma unroll
            for (int k = 0; k < 5; ++k) rec[k] = t[k];
        }
    }
}

__global__ void k_ordinal_depth_bwd(const float* __restrict__ d0, const float* __restrict__ d1,
                                    const float* __restrict__ a0, const float* __restrict__ a1,
                                    const unsigned char* __restrict__ m0, const unsigned char* __restrict__ m1, long n,
                                    const float* __restrict__ rec, const float* __restrict__ upstream,
                                    float* __restrict__ g0, float* __restrict__ g1, unsigned char* __restrict__ f0,
                                    unsigned char* __restrict__ f1)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;          // (with flags: n is a multiple of 64, so a wave is inside or outside as a whole)
    float r0 = 0.f, r1 = 0.f;
    if (a0[i] == 1.0f && a1[i] == 1.0f) {
        const float z0 = d0[i], z1 = d1[i];
        const bool b0 = m0[i] != 0, b1 = m1[i] != 0;
        const float up = upstream[0] / rec[0];
        if (b0 && !b1 && z1 < z0 && rec[1] > 0.f) {
            const float x = z0 - z1;
            if (x > 0.f && x < 2.f) { const float sg = hm_sigmoid(x); r0 += up * sg / rec[1]; r1 -= up * sg / rec[1]; }
        }
        if (b1 && !b0 && z0 < z1 && rec[3] > 0.f) {
            const float x = z1 - z0;
            if (x > 0.f && x < 2.f) { const float sg = hm_sigmoid(x); r1 += up * sg / rec[3]; r0 -= up * sg / rec[3]; }
        }
    }
    g0[i] = r0;
    g1[i] = r1;
    if (f0) {       // one byte per 64 consecutive pixels (= a segment of one row: S is a multiple of 64): "some gradient is non-zero"
        const unsigned long long b0 = __ballot(r0 != 0.f), b1 = __ballot(r1 != 0.f);
        if ((threadIdx.x & 63) == 0) { f0[i >> 6] = b0 ? 1 : 0; f1[i >> 6] = b1 ? 1 : 0; }
    }
}

extern "C" {
// Backward of the depth image of the last hm_sil_fwd (called with pooled_depth): grad_pooled_depth (B,S,S) ->
// grad_verts (B,V,3).  Uses faces9 / boxes / idx_map / owned of the workspace; gf9 scratch lives in `parts`.
// gflags (optional, S % 64 == 0): the non-zero structure of grad_pooled_depth as hm_ordinal_depth_bwd_flags leaves it - one byte per
// (frame, pixel row, 64-pixel segment), non-zero iff some pixel of the segment has a non-zero gradient (a byte may be set for an
// all-zero segment, never clear for a non-zero one).  Faces and frames that touch no flagged segment get their exact zeros
// without being walked; the result is the same with or without the flags.
int hm_depth_bwd_sparse(const float* verts, const float* K, int B, int V, int F, int S, float orig_size,
                        const float* grad_pooled_depth, const int* adj_off, const int* adj_items, float* grad_verts,
                        const unsigned char* gflags, void* workspace, hipStream_t stream)
{
    HM_CHECK_ARG(verts && K && grad_pooled_depth && adj_off && adj_items && grad_verts && workspace);
    HM_CHECK_ARG(!gflags || S % 64 == 0);
    if (S % 16 != 0) return HM_ERR_UNSUPPORTED;
    SilWs w = carve(workspace, B, V, F, S);
    hipLaunchKernelGGL(k_depth_bwd_faces, dim3(hm_cdiv(hm_cdiv((long)B * F, DBF_FACES) * 64, 256)), dim3(256), 0, stream, w.faces9, w.boxes,
                       w.idx_map, grad_pooled_depth, w.owned, B, F, S, (float*)w.parts, gflags);
    hipLaunchKernelGGL(k_depth_bwd_gather, dim3(hm_cdiv((long)B * V, 256)), dim3(256), 0, stream, (const float*)w.parts, adj_off,
                       adj_items, verts, K, B, V, F, orig_size, grad_verts, gflags, S);
    return hm_launch_status();
}
int hm_depth_bwd(const float* verts, const float* K, int B, int V, int F, int S, float orig_size,
                 const float* grad_pooled_depth, const int* adj_off, const int* adj_items, float* grad_verts,
                 void* workspace, hipStream_t stream)
{
    return hm_depth_bwd_sparse(verts, K, B, V, F, S, orig_size, grad_pooled_depth, adj_off, adj_items, grad_verts, nullptr, workspace,
                               stream);
}

// Ordinal depth loss between the object (layer 0) and the hand (layer 1).  workspace: hm_reduce_workspace_bytes()
// zero-filled once + B*8 floats of frame partials appended by the caller (frame_part).  rec (5) is kept for the backward.
int hm_ordinal_depth_fwd(const float* d0, const float* d1, const float* a0, const float* a1, const unsigned char* m0,
                         const unsigned char* m1, int B, int S, float* frame_part, float* rec, float* out1,
                         void* workspace, hipStream_t stream)
{
    HM_CHECK_ARG(d0 && d1 && a0 && a1 && m0 && m1 && frame_part && rec && out1 && workspace && B > 0 && S > 0);
    HM_CHECK_ARG(S <= 1024 && ((uintptr_t)frame_part & 7) == 0);
    hipLaunchKernelGGL(k_ordinal_depth, dim3(ORD_CHUNKS, B), dim3(256), 0, stream, d0, d1, a0, a1, m0, m1, B, S, frame_part,
                       (unsigned int*)((float*)workspace + 512), rec, out1);
    return hm_launch_status();
}
// flags0 / flags1 (both or neither; S % 64 == 0): B * S * (S / 64) bytes each, one per (frame, pixel row, 64-pixel segment), set to
// 1 where some pixel of the segment has a non-zero g0 / g1, to 0 elsewhere - the `gflags` of hm_depth_bwd_sparse.
int hm_ordinal_depth_bwd_flags(const float* d0, const float* d1, const float* a0, const float* a1, const unsigned char* m0,
                               const unsigned char* m1, int B, int S, const float* rec, const float* upstream, float* g0,
                               float* g1, unsigned char* flags0, unsigned char* flags1, hipStream_t stream)
{
    HM_CHECK_ARG(d0 && d1 && a0 && a1 && m0 && m1 && rec && upstream && g0 && g1);
    HM_CHECK_ARG((flags0 != nullptr) == (flags1 != nullptr) && (!flags0 || S % 64 == 0));
    const long n = (long)B * S * S;
    hipLaunchKernelGGL(k_ordinal_depth_bwd, dim3(hm_cdiv(n, 256)), dim3(256), 0, stream, d0, d1, a0, a1, m0, m1, n, rec,
                       upstream, g0, g1, flags0, flags1);
    return hm_launch_status();
}
int hm_ordinal_depth_bwd(const float* d0, const float* d1, const float* a0, const float* a1, const unsigned char* m0,
                         const unsigned char* m1, int B, int S, const float* rec, const float* upstream, float* g0,
                         float* g1, hipStream_t stream)
{
    return hm_ordinal_depth_bwd_flags(d0, d1, a0, a1, m0, m1, B, S, rec, upstream, g0, g1, nullptr, nullptr, stream);
}
}  // extern "C"
