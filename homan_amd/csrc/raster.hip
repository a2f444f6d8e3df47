// raster.hip -- NMR-semantics silhouette rasteriser for CDNA4 (gfx950), forward + pseudo-gradient.
//
// Replaces, on the reference's hot path, the third-party CUDA extension `neural_renderer`
// as called from reference homan/losses.py:187 (Renderer(...)(verts, faces, K=, mode="silhouettes"))
// with the ctor defaults of homan/losses.py:73-77 (anti_aliasing, fill_back, near=0.1, far=100,
// eps=1e-3), and fuses the masked-MSE / IoU reduction of homan/losses.py:188-197.
//
// Semantics (bit-compatible with oracle/csrc/nmr_raster.c given identical NDC faces):
//   hard z-buffer coverage on a (2S)^2 sample grid, inclusive edge test, perspective-correct
//   z from clamped+renormalised barycentrics, strict z-min with lowest-face-index tie-break,
//   fill_back (both windings, index f and F+f), vertical flip, 2x2 average pool; backward =
//   per-(face,edge,axis) line sweeps comparing in/out alpha (Kato et al. 2018).
//
// Design (CDNA4), see DESIGN.md section 4 for the measurements behind each choice:
//   forward   k_setup_faces (thread / face: optional rigid transform, projection, tight sample box, super-region bins; extra
//             workgroups write the camera-space vertices) -> k_raster_fwd (workgroup = 32x32-sample region: candidates of
//             its bin split into the camera-facing winding class and the hidden one; (candidate, 4x4 block) units
//             flattened over the threads; visibility by ds_min_u64 on an LDS z-buffer = the strict z test in ascending face
//             order, bit-exact; hidden-class units filtered against per-block depths and run on full waves of survivors;
//             epilogue per 8x8 output tile: index map, pooled silhouette, fused masked-MSE / IoU terms, sweep bit planes);
//   backward  k_bwd_lines (bit lines -> position-sorted source arrays + per-word records + per-line summaries; its first
//             workgroups build the flattened work list of the sweeps) -> k_bwd_sweep (persistent waves over 256-item
//             units: summary filter, then owner tests / slices / (item, source) pairs on full waves) -> per-corner NDC
//             gradients, gathered per vertex by k_bwd_gather or inside hm_rigid_bwd_sil.
#include "hm_common.h"
#include <type_traits>

#define HM_TILE 8          // output pixels per tile side
#define HM_STILE 16        // samples per tile side (2x SSAA)
#define RASTER_WAVES 4     // tiles per workgroup
#define STAGE_DW 20        // dwords per staged face in LDS (9 verts + 9 inverse + id + pad)

// output stores of the rasteriser's epilogue: -DRASTER_NT_STORES=1 makes them non-temporal (streamed past the L2: an A/B knob
// for the few microseconds of cache write-back between this launch and the next one of its chain)
typedef int hm_v2i __attribute__((ext_vector_type(2)));
typedef unsigned hm_v4u __attribute__((ext_vector_type(4)));
#ifndef RASTER_NT_STORES
#define RASTER_NT_STORES 0
#endif
__device__ __forceinline__ void hm_out_store(float* p, float v)
{
#if RASTER_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void hm_out_store2(int* p, int a, int b)
{
#if RASTER_NT_STORES
    hm_v2i v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<hm_v2i*>(p));
#else
    *reinterpret_cast<int2*>(p) = make_int2(a, b);
#endif
}
__device__ __forceinline__ void hm_out_store4(unsigned short* p, unsigned a, unsigned b, unsigned c, unsigned d)
{
#if RASTER_NT_STORES
    hm_v4u v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<hm_v4u*>(p));
#else
    *reinterpret_cast<uint4*>(p) = make_uint4(a, b, c, d);
#endif
}

struct FaceBox {           // 8 bytes per face: sample-space box + winding mask in x0[15:14]
    unsigned short x0m, y0, x1, y1;
};

__device__ __forceinline__ float topix(float v, int is)
{
    float a = v * (float)is;
    a = a + (float)is;
    a = a - 1.0f;
    return 0.5f * a;
}
__device__ __forceinline__ bool backside(const float* f)
{
    return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);
}
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------- projection (nr.projection, zero distortion)
// camera-space vertex, K (3,3) -> (u, v, z), u,v in [-1,1], v up.
__device__ __forceinline__ void project_vertex(const float* __restrict__ p, const float* __restrict__ k, float orig_size,
                                               float* out)
{
    const float x = p[0], y = p[1], z = p[2];
    const float zz = z + 1e-9f;
    const float xn = x / zz, yn = y / zz;
    float u = xn * k[0] + yn * k[1];
    u = u + k[2];
    float v = xn * k[3] + yn * k[4];
    v = v + k[5];
    v = orig_size - v;
    u = 2.0f * (u - orig_size / 2.0f) / orig_size;
    v = 2.0f * (v - orig_size / 2.0f) / orig_size;
    out[0] = u; out[1] = v; out[2] = z;
}

// ---------------------------------------------------------------- face setup
// projects the three vertices of every face (a vertex is shared by ~6 faces: re-projecting it is cheaper than a
// separate projection launch on the critical path), packs the (B,F,3,3) NDC face buffer and the 8-byte screen boxes.
// It also bins the faces into super-regions of 64x64 or 128x128 samples (hm_sr_shift) (counts aggregated per workgroup in LDS, one global atomic per
// (workgroup, bin)): a raster workgroup then scans the faces of its super-region instead of the whole frame.  The order
// inside a bin is arbitrary; the raster resolves visibility with a min, so its result does not depend on it.
// grid (ceil(F/256), B).  bin_cnt must be zero on entry (the raster's last workgroup resets it).
#define SR_MAX 64          // super-regions per frame (is <= 1024)
// log2 of the super-region side in samples: 64^2 up to 512^2 samples, 128^2 above -> at most 8 x 8 = SR_MAX bins.  A
// region workgroup scans its whole bin, so a bin holds 4 (is <= 512) or 16 regions' worth of faces: the finer bins cut
// the scan of the 512^2 rasters by 4 (k_raster_fwd 63.5 -> 60.1 us).
__host__ __device__ __forceinline__ int hm_sr_shift(int is) { return is <= 512 ? 6 : 7; }
__global__ __launch_bounds__(256) void k_setup_faces(const float* __restrict__ verts, const float* __restrict__ K,
                                                     float orig_size, const int* __restrict__ faces, int faces_bstride,
                                                     int B, int V, int F, int is, float* __restrict__ faces9,
                                                     FaceBox* __restrict__ boxes, unsigned char* __restrict__ owned,
                                                     int* __restrict__ bin_cnt, int* __restrict__ bin_list,
                                                     const float* __restrict__ rigid_rot6d,
                                                     const float* __restrict__ rigid_trans,
                                                     const float* __restrict__ rigid_scale, int rigid_abs, int clip_len,
                                                     float* __restrict__ cam_out, int nfb)
{
    HM_STAMP_START(0);
    __shared__ int s_cnt[SR_MAX], s_base[SR_MAX];
    __shared__ float s_R[9];
    // optional rigid transform of mesh-space `verts` (same arithmetic as hm_rigid_fwd, so the camera-space vertices the
    // other losses get from that entry point are the very numbers rasterised here): the silhouette chain then does not
    // wait for a separate transform launch
    if (rigid_rot6d && threadIdx.x == 0) rot6d_to_mat(rigid_rot6d + blockIdx.y * 6, s_R);
    if ((int)blockIdx.x >= nfb) {
        // vertex blocks behind the face blocks (cam_out != NULL): the camera-space vertices themselves, for the caller's
        // other losses - the arithmetic of k_rigid_fwd, so hm_rigid_fwd on the same inputs returns the same floats, and the
        // caller's second stream no longer opens with a transform launch of its own
        __syncthreads();
        const int bb = blockIdx.y, v = ((int)blockIdx.x - nfb) * blockDim.x + threadIdx.x;
        if (v >= V) return;
        float sc = rigid_scale[bb / clip_len];
        if (rigid_abs) sc = fabsf(sc);
        const float* m = verts + ((long)bb * V + v) * 3;
        const float x = sc * m[0], y = sc * m[1], z = sc * m[2];
        const float* t = rigid_trans + bb * 3;
        float* o = cam_out + ((long)bb * V + v) * 3;
        o[0] = x * s_R[0] + y * s_R[3] + z * s_R[6] + t[0];
        o[1] = x * s_R[1] + y * s_R[4] + z * s_R[7] + t[1];
        o[2] = x * s_R[2] + y * s_R[5] + z * s_R[8] + t[2];
        return;
    }
    const int b = blockIdx.y, fi = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = fi < F;
    const int nsx = (is + (1 << hm_sr_shift(is)) - 1) >> hm_sr_shift(is), nsr = nsx * nsx;
    if (threadIdx.x < SR_MAX) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    unsigned mask = 0;
    int x0 = 1, y0 = 1, x1 = 0, y1 = 0;
    if (valid) {
        const long i = (long)b * F + fi;
        const int* fc = faces + (long)b * faces_bstride + 3 * fi;
        float f[9], r[9];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* mv = verts + ((long)b * V + fc[k]) * 3;
            float cam[3] = {mv[0], mv[1], mv[2]};
            if (rigid_rot6d) {
                float sc = rigid_scale[b / clip_len];      // one object scale per clip
                if (rigid_abs) sc = fabsf(sc);
                const float x = sc * mv[0], y = sc * mv[1], z = sc * mv[2];
                const float* t = rigid_trans + b * 3;
                cam[0] = x * s_R[0] + y * s_R[3] + z * s_R[6] + t[0];
                cam[1] = x * s_R[1] + y * s_R[4] + z * s_R[7] + t[1];
                cam[2] = x * s_R[2] + y * s_R[5] + z * s_R[8] + t[2];
            }
            project_vertex(cam, K + b * 9, orig_size, f + 3 * k);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { r[3 * k] = f[3 * (2 - k)]; r[3 * k + 1] = f[3 * (2 - k) + 1]; r[3 * k + 2] = f[3 * (2 - k) + 2]; }
#pragma unroll
        for (int k = 0; k < 9; ++k) faces9[i * 9 + k] = f[k];
        owned[(long)b * 2 * F + fi] = 0;          // "owns at least one sample" flags, set by the forward raster
        owned[(long)b * 2 * F + F + fi] = 0;
        mask = (backside(f) ? 0u : 1u) | (backside(r) ? 0u : 2u);
        float px[3], py[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { px[k] = topix(f[3 * k], is); py[k] = topix(f[3 * k + 1], is); }
        const float xmin = fminf(px[0], fminf(px[1], px[2])), xmax = fmaxf(px[0], fmaxf(px[1], px[2]));
        const float ymin = fminf(py[0], fminf(py[1], py[2])), ymax = fmaxf(py[0], fmaxf(py[1], py[2]));
        if (!(xmax >= -2.0f && ymax >= -2.0f && xmin <= is + 1.0f && ymin <= is + 1.0f)) mask = 0;  // off-screen / NaN
        // a vertex projected beyond 1e15 (a camera-space depth within 1e-15 of the image plane), or not finite: the edge
        // functions of such a face overflow to inf - inf; culled here and in the oracle (oracle/csrc/nmr_raster.c)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (!(fabsf(f[3 * k]) <= 1e15f && fabsf(f[3 * k + 1]) <= 1e15f)) mask = 0;
        // sample p (integer pixel coordinate) can be covered only if min <= p <= max; 0.01 px of slack dwarfs the
        // rounding of the edge functions, so the box is conservative yet tight
        x0 = max(0, (int)ceilf(fmaxf(xmin, -2.0f) - 0.01f));
        x1 = min(is - 1, (int)floorf(fminf(xmax, is + 1.0f) + 0.01f));
        y0 = max(0, (int)ceilf(fmaxf(ymin, -2.0f) - 0.01f));
        y1 = min(is - 1, (int)floorf(fminf(ymax, is + 1.0f) + 0.01f));
        if (x1 < x0 || y1 < y0) mask = 0;
        if (mask == 0) { x0 = y0 = 1; x1 = y1 = 0; }
        FaceBox bx;
        bx.x0m = (unsigned short)(x0 | (mask << 14));
        bx.y0 = (unsigned short)y0;
        bx.x1 = (unsigned short)x1;
        bx.y1 = (unsigned short)y1;
        boxes[i] = bx;
    }
    if (!bin_cnt) return;
    // local slots in LDS, one global reservation per (workgroup, bin)
    const int sx0 = x0 >> hm_sr_shift(is), sx1 = x1 >> hm_sr_shift(is), sy0 = y0 >> hm_sr_shift(is), sy1 = y1 >> hm_sr_shift(is);
    int local[4];                       // a face larger than 2x2 super-regions reserves its further bins one by one
    int nloc = 0;
    if (mask)
        for (int sy = sy0; sy <= sy1; ++sy)
            for (int sx = sx0; sx <= sx1; ++sx) {
                if (nloc < 4) local[nloc] = atomicAdd(&s_cnt[sy * nsx + sx], 1);
                ++nloc;
            }
    __syncthreads();
    if (threadIdx.x < nsr) {
        const int c = s_cnt[threadIdx.x];
        s_base[threadIdx.x] = c ? atomicAdd(&bin_cnt[b * nsr + threadIdx.x], c) : 0;
    }
    __syncthreads();
    if (mask) {
        int q = 0;
        for (int sy = sy0; sy <= sy1; ++sy)
            for (int sx = sx0; sx <= sx1; ++sx, ++q) {
                const int sr = sy * nsx + sx;
                const int at = q < 4 ? s_base[sr] + local[q] : atomicAdd(&bin_cnt[b * nsr + sr], 1);
                bin_list[((long)b * nsr + sr) * F + at] = fi;
            }
    }
    HM_STAMP_END(0);
}

__device__ __forceinline__ int hm_wave_scan_incl(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);    // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);    // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ float rlane(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// Sweep bit planes of one 8x8-pixel tile (wave = tile, lane = output pixel (lane>>3, lane&7), bal[2*dy+dx] = ballot of
// "sample (dy,dx) of my pixel is covered"): plane 0 = samples with alpha==0 and g<0 ("wants to be filled", walked by the
// outward sweeps), plane 1 = samples with alpha==1 and g>0 ("wants to be emptied", the only samples the inward sweeps
// can collect from); each plane as row words (bits along x) and column words (bits along y).
__device__ __forceinline__ unsigned spread8(unsigned x)      // bit k of x -> bit 2k
{
    x = (x | (x << 4)) & 0x0f0fu;
    x = (x | (x << 2)) & 0x3333u;
    x = (x | (x << 1)) & 0x5555u;
    return x;
}
// Layout of the sweep planes: TILE-BLOCKED, (B, T, T, 4, 16) u16 with T = is/16 tiles per side and the four
// (orientation, plane) combinations of a tile side by side: [row words plane 0 | row words plane 1 | column words plane 0 |
// column words plane 1], 16 words each = the tile's 128 bytes, written by its wave as ONE full cache line.  (Line-major
// planes made every 2-byte word of a tile a partial write into a different line, and since neighbouring tiles run on
// different XCDs their dirty fragments never merged in an L2: 64 partial HBM writes per tile, 140 MB per launch.)
// A 64-sample word of a line is four tiles' words (hm_plane_word64).
__device__ __forceinline__ long hm_plane_at(int b, int ty, int tx, int combo, int T)
{
    return ((((long)b * T + ty) * T + tx) * 4 + combo) * 16;
}
// 64 samples [64k, 64k+64) of line d0 of `axis` (1: sample row yi = d0, bits along x ; 0: sample column xi = d0, bits
// along the 16-sample groups ygrp of emit_planes), plane pl
__device__ __forceinline__ unsigned long long hm_plane_word64(const unsigned short* __restrict__ planes, int b, int is,
                                                              int axis, int pl, int d0, int k)
{
    const int T = is / 16;
    unsigned long long w = 0ull;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned v;
        if (axis) {
            const int t = is - 1 - d0;
            v = planes[hm_plane_at(b, t >> 4, 4 * k + j, pl, T) + (t & 15)];
        } else {
            v = planes[hm_plane_at(b, T - 1 - (4 * k + j), d0 >> 4, 2 + pl, T) + (d0 & 15)];
        }
        w |= (unsigned long long)v << (16 * j);
    }
    return w;
}
// nb / pb [2*dy+dx] = ballots of "g < 0" / "g > 0" per sample (all four equal when g lives on the pooled grid).
struct Ballots4 { unsigned long long s0, s1, s2, s3; };      // [2*dy+dx]; plain scalars, picked with selects only:
// a lane-dependent index into an array puts the array in scratch memory (80-112 bytes per thread = 100 MB of HBM writes
// per launch when this was `cov[2 * sub]`)
__device__ __forceinline__ void emit_planes(const Ballots4 cov, const Ballots4 nb, const Ballots4 pb, int b, int B, int is,
                                            int tx, int ty, int lane, unsigned short* __restrict__ planes)
{
    // lanes 0..15: row word of sample row (rr2 = l>>1, dy = l&1), plane 0 ; lanes 16..31: same for plane 1 ;
    // lanes 32..47: column word of sample column (cc2 = l>>1, dx = l&1), plane 0 ; lanes 48..63: plane 1
    const int l = lane & 15, pl = (lane >> 4) & 1, hi = l >> 1, sub = l & 1;
    unsigned long long b0, b1;          // the two ballots this lane interleaves
    unsigned word;
    const Ballots4 g = pl == 0 ? nb : pb;
    if (lane < 32) {                    // (dy = sub): dx = 0 -> even bits, dx = 1 -> odd bits ; sample row
        const unsigned long long c0 = sub ? cov.s2 : cov.s0, c1 = sub ? cov.s3 : cov.s1;      // yi = is-1-2*(ty*8+hi)-sub,
        const unsigned long long g0 = sub ? g.s2 : g.s0, g1 = sub ? g.s3 : g.s1;              // i.e. (is-1-yi) & 15 = l
        b0 = pl == 0 ? (~c0 & g0) : (c0 & g0);
        b1 = pl == 0 ? (~c1 & g1) : (c1 & g1);
        const unsigned a = (unsigned)(b0 >> (8 * hi)) & 0xffu, o = (unsigned)(b1 >> (8 * hi)) & 0xffu;
        word = spread8(a) | (spread8(o) << 1);
    } else {                            // (dx = sub): bit 15 - (2*rr2 + dy) <- sample (rr2, dy) of column cc2 = hi
        const unsigned long long c0 = sub ? cov.s1 : cov.s0, c1 = sub ? cov.s3 : cov.s2;      // dy = 0 / dy = 1 ; sample column
        const unsigned long long g0 = sub ? g.s1 : g.s0, g1 = sub ? g.s3 : g.s2;              // xi = 16*tx + l
        b0 = pl == 0 ? (~c0 & g0) : (c0 & g0);
        b1 = pl == 0 ? (~c1 & g1) : (c1 & g1);
        // bits 8*rr2 + cc2 -> one byte with row rr2 at bit 7 - rr2
        const unsigned k0 = (unsigned)((((b0 >> hi) & 0x0101010101010101ull) * 0x8040201008040201ull) >> 56);
        const unsigned k1 = (unsigned)((((b1 >> hi) & 0x0101010101010101ull) * 0x8040201008040201ull) >> 56);
        word = (spread8(k0) << 1) | spread8(k1);    // the word of the 16-sample group ygrp = T-1-ty along y
    }
    // the 64 words of the tile (lane order = memory order, combo = lane >> 4) leave as eight 16-byte stores: sub-dword
    // stores reach HBM as one partial write each (measured: 54 B of WRITE_SIZE per 2-byte store)
    const unsigned p2 = (word & 0xffffu) | ((unsigned)__builtin_amdgcn_update_dpp(0, (int)word, 0x101, 0xf, 0xf, false) << 16);   // row_shl:1
    const unsigned q2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p2, 0x102, 0xf, 0xf, false);      // row_shl:2 : words 2,3
    const unsigned r2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p2, 0x104, 0xf, 0xf, false);      // row_shl:4 : words 4,5
    const unsigned s2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)q2, 0x104, 0xf, 0xf, false);      // row_shl:4 : words 6,7
    if ((lane & 7) == 0) hm_out_store4(planes + hm_plane_at(b, ty, tx, 0, is / 16) + lane, p2, q2, r2, s2);
}

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
// In-graph timing (hm_sil_timestamps): byte 1 of workspace word 24 switches it on; then every workgroup of the three heavy
// kernels (every wave of the persistent sweep kernel) stores the device wall clock (s_memrealtime, constant rate) at its
// entry and at its exit into a slot pair of its own - plain 8-byte stores, no same-address atomics that would stretch the
// kernel being measured; the host takes the earliest start and the latest end.  Off: one scalar load per workgroup.
__device__ __forceinline__ bool hm_ts_enabled(const unsigned int* __restrict__ flag_word)
{
    return flag_word && ((flag_word[0] >> 8) & 1u);
}
__device__ __forceinline__ void hm_ts_store(unsigned long long* __restrict__ slots, long unit, int which, unsigned long long t)
{
    __builtin_nontemporal_store(t, slots + 2 * unit + which);
}
#ifdef RASTER_PHASES
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
__global__ __launch_bounds__(64 * RASTER_WAVES) __attribute__((amdgpu_waves_per_eu(RASTER_WPE, 8))) void k_raster_fwd(
    const float* __restrict__ faces9, const FaceBox* __restrict__ boxes, int B, int F, int S, float znear,
    float zfar, int* __restrict__ idx_map, unsigned short* __restrict__ alpha16, float* __restrict__ pooled,
    const float* __restrict__ keep, const float* __restrict__ ref, float* __restrict__ dimg,
    float* __restrict__ partials, const int* __restrict__ work_order, unsigned char* __restrict__ owned,
    float* __restrict__ pooled_depth, unsigned short* __restrict__ planes,
    int* __restrict__ bin_cnt, const int* __restrict__ bin_list, unsigned int* __restrict__ done, int reset_bins,
    unsigned char* __restrict__ region_state, int persistent, float* __restrict__ alpha_full, int mask_shared,
    float* __restrict__ dimg_full, const unsigned int* __restrict__ hint, unsigned long long* __restrict__ ts_slots,
    int* __restrict__ wo_dyn, unsigned int* __restrict__ wg_cost)
{
    HM_CHAIN_KERNEL();
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
    const int wo = __builtin_amdgcn_readfirstlane(dyn ? wo_dyn[blockIdx.x]
                                                     : work_order ? work_order[blockIdx.x] : (int)(((blockIdx.x % B) << 16) | (blockIdx.x / B)));
    if (wo_dyn && !dyn && threadIdx.x == 0) wo_dyn[blockIdx.x] = wo;
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
    if (ts_on) hm_ts_store(ts_slots, blockIdx.x, 0, ts_t0);
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
        if (wg_cost && tid == 0) wg_cost[blockIdx.x] = 0u;
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
    if (ts_on) hm_ts_store(ts_slots, blockIdx.x, 1, (unsigned long long)wall_clock64());
    if (wg_cost && tid == 0) wg_cost[blockIdx.x] = (unsigned)min((unsigned long long)wall_clock64() - ts_t0, 0xfffffffeull) + 1u;
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

// grid (B): per-frame sums of the tile partials, then the last block of every clip (clip_len consecutive frames) finishes:
// loss = (sum_sq / keep_sum[clip]) / clip_len ; iou = mean_b inter_b / (union_b + eps) over the clip's frames.
// out[clip*out_stride + 0]=loss, [+1]=iou.  The clip's ticket word is slot 3 of the frame record of its first frame.
__device__ __forceinline__ void sil_reduce_frame(int b, const float* __restrict__ partials, int ntiles,
                                                 const float* __restrict__ keep_sum, float* __restrict__ frame_rec,
                                                 float* __restrict__ out, float* __restrict__ frame_out, int clip_len,
                                                 int out_stride)
{
    __shared__ float red[16];
    __shared__ int s_flag;
    float sq = 0.f, in = 0.f, un = 0.f;
    for (int t = threadIdx.x; t < ntiles; t += blockDim.x) {
        const float4 p = *reinterpret_cast<const float4*>(partials + ((long)b * ntiles + t) * 4);
        sq += p.x; in += p.y; un += p.z;
    }
    sq = hm_block_sum(sq, red);
    in = hm_block_sum(in, red);
    un = hm_block_sum(un, red);
    if (threadIdx.x == 0) { hm_partial_store(frame_rec + 4 * b, sq); hm_partial_store(frame_rec + 4 * b + 1, in / (un + 1e-6f)); }
    // per-frame values (un-normalised sum of squares, IoU): what a loss that keeps the frames apart needs
    if (frame_out && threadIdx.x == 0) { frame_out[2 * b] = sq; frame_out[2 * b + 1] = in / (un + 1e-6f); }
    if (!out) return;
    const int clip = b / clip_len;
    float* crec = frame_rec + 4L * clip * clip_len;
    if (hm_last_block(reinterpret_cast<unsigned int*>(crec + 3), clip_len, &s_flag)) {
        const float total_sq = hm_last_block_sum(crec, clip_len, 4, red);
        const float iou_sum = hm_last_block_sum(crec + 1, clip_len, 4, red);
        if (threadIdx.x == 0) {
            out[(long)clip * out_stride] = (total_sq / keep_sum[clip]) / (float)clip_len;
            out[(long)clip * out_stride + 1] = iou_sum / (float)clip_len;
        }
    }
}
__global__ __launch_bounds__(256) void k_sil_reduce(const float* __restrict__ partials, int B, int ntiles,
                                                     const float* __restrict__ keep_sum, float* __restrict__ frame_rec,
                                                     float* __restrict__ out, float* __restrict__ frame_out, int clip_len,
                                                     int out_stride)
{
    HM_LATENCY_KERNEL();
    sil_reduce_frame(blockIdx.x, partials, ntiles, keep_sum, frame_rec, out, frame_out, clip_len, out_stride);
}

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
struct SweepFace {            // 64 B: one face of the flattened work list
    int bf, b, off, flags;    // face slot b*F+fi, frame, first item, bit 0: accumulate with atomics (capacity overflow)
    float px[3], py[3];       // pixel-space corners (px[0..2], py[0..2] contiguous)
    unsigned short cum[12];   // inclusive item counts of the families, family = winding*6 + edge*2 + axis
};
#define SWEEP_PASS_FACES 16   // faces of a unit staged in LDS at a time (a unit with more takes several passes)
#ifndef SWEEP_USHIFT
#define SWEEP_USHIFT 8         // a unit = 256 consecutive items of the global list (see k_bwd_sweep)
#endif
#define SWEEP_UNIT (1 << SWEEP_USHIFT)
#define SWEEP_TRIPS (SWEEP_UNIT / 64)      // stage-1 trips of a unit
#define SWEEP_TBATCH (SWEEP_TRIPS < 4 ? SWEEP_TRIPS : 4)      // trips whose loads are in flight together
struct SweepItem { float x, c0, c1; int base0, base1, nb0, fn, meta; };   // meta: face | t0<<4 | t1<<7 | use0<<10 | use1<<11
struct SweepList {
    SweepFace* tab; int* offs; unsigned int* ufirst; unsigned int* tickets; float* upart;
    unsigned long long* cnt; unsigned int* done; unsigned long long* total;
    int ucap, slot_cap;
};

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
struct SweepSrc { int d1; float g; int owner; };
#define SWEEP_CUMW 16           // cumulative-count slots per line (is <= 1024)

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

// Global addresses of the hot loops: with W32 (every array of the workspace below 4 GB, checked by the host) an element address
// is the array's base - a scalar register pair - plus a 32-bit BYTE offset formed in 32-bit arithmetic, which is the
// addressing mode of the global loads themselves; with 64-bit element indices a third of stage 1's instructions were the
// 64-bit multiply-adds, sign extensions and shifts of its three addresses.
template <bool W32> struct HmOff { typedef long t; };
template <> struct HmOff<true> { typedef unsigned t; };
template <bool W32, class T>
__device__ __forceinline__ T* hm_at(T* __restrict__ base, typename HmOff<W32>::t i)          // (T may be const)
{
    typedef typename std::conditional<std::is_const<T>::value, const char, char>::type byte_t;
    if (W32) return reinterpret_cast<T*>(reinterpret_cast<byte_t*>(base) + (unsigned)(i * (unsigned)sizeof(T)));
    return base + i;
}

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
#if defined(SWEEP_EXP) && SWEEP_EXP == 2          // (timing experiment: metadata loads only; results are wrong)
            if (eps > 0.f) { if (nfp < SWEEP_PASS_FACES) break; else continue; }
#endif
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
#if defined(SWEEP_EXP) && SWEEP_EXP == 1          // (timing experiment: no stage 2; results are wrong)
            if (eps > 0.f) qn = 0;
#endif
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
#if defined(SWEEP_EXP) && SWEEP_EXP == 3          // (timing experiment: at most one pair round per trip; results are wrong)
                    if (base > 0 && eps > 0.f) break;
#endif
#if defined(SWEEP_EXP) && SWEEP_EXP == 4          // (timing experiment: no pair rounds at all; results are wrong)
                    if (eps > 0.f) break;
#endif
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
#if defined(SWEEP_EXP) && SWEEP_EXP == 5          // (timing experiment: every lane its own accumulator - no same-address atomics; results are wrong)
                                    double* f = s_fg[wv][lane & 15];
                                    unsafeAtomicAdd(f + (lane >> 4), acc0);
                                    unsafeAtomicAdd(f + 4 + ((lane >> 4) & 1), acc1);
#elif defined(SWEEP_EXP) && SWEEP_EXP == 6        // (timing experiment: no LDS atomics in the pair loop; results are wrong)
                                    if (eps < 0.f) s_fg[wv][0][0] = acc0 + acc1;
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
                                                         float* __restrict__ gf9)
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
                                   float* __restrict__ grad_verts)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
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
//   words 2-3  pixels ordered wrongly: (annotated 0 in front) | (annotated 1 in front) << 32
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
    __shared__ float red[16];
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
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = hm_block_sum(v[k], red);
    unsigned long long* fr = reinterpret_cast<unsigned long long*>(frame_part) + (long)b * 4;
    if (threadIdx.x == 0) {
        atomicAdd(fr, (unsigned long long)v[0] | ((unsigned long long)v[1] << 21) | ((unsigned long long)v[2] << 42));
        atomicAdd(fr + 1, (unsigned long long)v[3] | ((unsigned long long)v[5] << 32));
        atomicAdd(fr + 2, (unsigned long long)((double)v[4] * ORD_FIX));
        atomicAdd(fr + 3, (unsigned long long)((double)v[6] * ORD_FIX));
    }
    if (hm_last_block(counter, gridDim.x * gridDim.y, &s_flag)) {
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
                t[1] += (float)(unsigned)(w1 & 0xffffffffull);
                t[2] += (float)((double)w2 / ORD_FIX);
                t[3] += (float)(unsigned)(w1 >> 32);
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
                                    float* __restrict__ g0, float* __restrict__ g1)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
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
}

// ================================================================ C ABI
// persistent sweep waves: 4 per SIMD.  More does not speed the sweep up and starves the concurrent hand-side kernels
// of wave slots (they run on a second stream of the same hipGraph).
#ifndef SWEEP_BLOCKS
#define SWEEP_BLOCKS 1280
#endif
// capacity of the sweep work list: units (64 items) indexed by `ufirst`, and per-unit partial slots of faces spread over
// several units.  256 items per face on average is ~5x what a mesh filling the image produces; beyond it the sweep stays
// correct (binary search for a unit's first face, float atomics for the faces past the slot table), only slower.
static int g_sweep_cap_override = 0;
// measurement hook (hm_debug_sil_timing): HIP events recorded on the launch stream right before / after the three heavy
// kernels of the silhouette chain, so a caller that drives the optimisation loop launch by launch (not from a captured
// graph) reads the duration each kernel had INSIDE the loop, next to whatever runs on the other streams
static hipEvent_t g_tev[6];
static int g_timing = 0;
#define HM_TIME_MARK(k, stream) do { if (g_timing) (void)hipEventRecord(g_tev[k], stream); } while (0)
// timestamp slots of hm_sil_timestamps: {start, end} per raster workgroup, per lines workgroup, per sweep wave
#define TS_SWEEP_WGS 4096
static inline size_t ts_raster_units(int B, int S) { return (size_t)B * (S / 8) * (S / 8) / RASTER_WAVES; }
static inline size_t ts_lines_units(int B, int F, int S) { return (size_t)B * F / 256 + 2 * (size_t)B + 64 + (size_t)B * S / 2 + 8; }
static inline size_t ts_units(int B, int F, int S) { return ts_raster_units(B, S) + ts_lines_units(B, F, S) + 4 * TS_SWEEP_WGS; }
static inline size_t sweep_ucap(int B, int F) { return (size_t)B * F * 4 + 1024; }
static inline size_t sweep_slot_cap(int B, int F) { return sweep_ucap(B, F) + (size_t)B * F; }
extern "C" {

// workspace layout helper (bytes), all chunks 256-byte aligned
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t hm_sil_workspace_bytes(int B, int V, int F, int S)
{
    const size_t is = 2 * (size_t)S;
    size_t n = 256;                             // counter word (zero-initialised by the caller once)
    n += al256((size_t)B * 16);                 // per-frame reduction records
    n += al256((size_t)B * V * 3 * 4);          // ndc
    n += al256((size_t)B * F * 9 * 4);          // faces9
    n += al256((size_t)B * F * 8);              // boxes
    n += al256((size_t)B * is * is * 4);        // idx_map
    n += al256((size_t)B * is * (is / 16) * 2); // alpha16
    n += al256((size_t)B * S * S * 4);          // dimg
    n += al256((size_t)B * (S / 8) * (S / 8) * 16); // partials
    n += al256((size_t)B * is * is * 4);        // gimg (pooled grid, or the full sample grid without anti-aliasing)
    n += al256((size_t)B * is * (is / 16) * 4); // row masks, 2 planes
    n += al256((size_t)B * is * (is / 16) * 4); // column masks, 2 planes
    n += al256((size_t)B * F * 24 * 4);         // parts
    n += al256((size_t)B * F * 2);              // owned
    n += al256((size_t)B * SR_MAX * 8);                        // super-region bin counters + their ticket words
    n += al256((size_t)B * (S / 16) * (S / 16));               // per-region "outputs hold the empty pattern" flags
    n += al256((size_t)B * SR_MAX * F * 4);                    // super-region face lists (worst case: every face in every bin)
    n += al256(4 * (size_t)B * is * (is / 64) * 16);            // per-line records {mask word, sources before it}
    n += al256((size_t)B * 2 * is * 16);                        // per-line summaries {first, last+1, word mask} x 2 planes
    n += al256(4 * (size_t)B * is * is * sizeof(SweepSrc));     // per-line source arrays (2 planes x 2 orientations)
    n += al256((size_t)B * F * sizeof(SweepFace));              // sweep work list: face records,
    n += al256((size_t)B * F * 4) * 2;                          //   their first items, their tickets,
    n += al256(sweep_ucap(B, F) * 4);                           //   first face of every unit,
    n += al256(sweep_slot_cap(B, F) * 24);                      //   per-unit partials of faces spread over several units
    n += al256(ts_units(B, F, S) * 16);                          // in-graph timestamps (hm_sil_timestamps)
    n += al256(ts_raster_units(B, S) * 4) * 3;                   // adaptive raster launch order: entries, scratch, times
    return n;
}

struct SilWs {
    unsigned int* counter; float* frame_rec;
    float* ndc; float* faces9; FaceBox* boxes; int* idx_map; unsigned short* alpha16; float* dimg;
    float* partials; float* gimg; unsigned short* planes; double* parts;
    unsigned char* owned; int* bin_cnt; unsigned int* bin_done; unsigned char* region_state; int* bin_list;
    uint4* lrec; SweepSrc* srcs; unsigned short* lsum;
    SweepList sweep;
    unsigned long long* ts;      // {start, end} slots: raster workgroups | lines workgroups | sweep waves
    int* wo_dyn; int* wo_tmp; unsigned int* wg_cost;      // adaptive raster launch order (hm_tune_raster_reorder)
};
static SilWs carve(void* ws, int B, int V, int F, int S)
{
    const size_t is = 2 * (size_t)S;
    char* p = (char*)ws;
    SilWs w;
    w.counter = (unsigned int*)p; p += 256;
    w.frame_rec = (float*)p; p += al256((size_t)B * 16);
    w.ndc = (float*)p; p += al256((size_t)B * V * 3 * 4);
    w.faces9 = (float*)p; p += al256((size_t)B * F * 9 * 4);
    w.boxes = (FaceBox*)p; p += al256((size_t)B * F * 8);
    w.idx_map = (int*)p; p += al256((size_t)B * is * is * 4);
    w.alpha16 = (unsigned short*)p; p += al256((size_t)B * is * (is / 16) * 2);
    w.dimg = (float*)p; p += al256((size_t)B * S * S * 4);
    w.partials = (float*)p; p += al256((size_t)B * (S / 8) * (S / 8) * 16);
    w.gimg = (float*)p; p += al256((size_t)B * is * is * 4);
    w.planes = (unsigned short*)p; p += al256((size_t)B * is * (is / 16) * 4) * 2;      // (B, T, T, 4, 16) u16
    w.parts = (double*)p; p += al256((size_t)B * F * 24 * 4);      // (6 doubles per face for the sweeps; 9 floats for the depth backward)
    w.owned = (unsigned char*)p; p += al256((size_t)B * F * 2);
    w.bin_cnt = (int*)p; w.bin_done = (unsigned int*)(p + (size_t)B * SR_MAX * 4); p += al256((size_t)B * SR_MAX * 8);
    w.region_state = (unsigned char*)p; p += al256((size_t)B * (S / 16) * (S / 16));
    w.bin_list = (int*)p; p += al256((size_t)B * SR_MAX * F * 4);
    w.lrec = (uint4*)p; p += al256(4 * (size_t)B * is * (is / 64) * 16);
    w.lsum = (unsigned short*)p; p += al256((size_t)B * 2 * is * 16);
    w.srcs = (SweepSrc*)p; p += al256(4 * (size_t)B * is * is * sizeof(SweepSrc));
    w.sweep.tab = (SweepFace*)p; p += al256((size_t)B * F * sizeof(SweepFace));
    w.sweep.offs = (int*)p; p += al256((size_t)B * F * 4);
    w.sweep.tickets = (unsigned int*)p; p += al256((size_t)B * F * 4);
    w.sweep.ufirst = (unsigned int*)p; p += al256(sweep_ucap(B, F) * 4);
    w.sweep.upart = (float*)p; p += al256(sweep_slot_cap(B, F) * 24);
    w.ts = (unsigned long long*)p; p += al256(ts_units(B, F, S) * 16);
    w.wo_dyn = (int*)p; p += al256(ts_raster_units(B, S) * 4);
    w.wo_tmp = (int*)p; p += al256(ts_raster_units(B, S) * 4);
    w.wg_cost = (unsigned int*)p;
    w.sweep.cnt = (unsigned long long*)(w.counter + 16);        // zero between launches (re-armed by the last compaction block)
    w.sweep.done = w.counter + 18;
    w.sweep.total = (unsigned long long*)(w.counter + 20);
    w.sweep.ucap = (int)sweep_ucap(B, F);
    w.sweep.slot_cap = (int)sweep_slot_cap(B, F);
    if (g_sweep_cap_override > 0) {          // test hook (hm_debug_sweep_caps): force the beyond-capacity paths
        w.sweep.ucap = min(w.sweep.ucap, g_sweep_cap_override);
        w.sweep.slot_cap = min(w.sweep.slot_cap, g_sweep_cap_override);
    }
    return w;
}

// Scheduling hint, no effect on results (hm_tune_raster_reorder): while it is on, the forward launches of hm_sil_fwd record
// what every raster workgroup cost and the first launch of hm_sil_bwd re-sorts the raster's launch order by it (kept in the
// workspace; the caller's work_order only seeds it).  Read when the entry points are called (or captured).
static thread_local int g_raster_reorder = 0;
// pass 2a (+ the work list of pass 2b in its first workgroups) and pass 2b
// 32-bit byte offsets (the W32 instantiations of the line expansion and the sweeps) while the largest array they index - the
// per-line source arrays, 4 B is^2 records of 12 bytes - stays below 4 GB.  HOMAN_FORCE_W64=1 (read once; a test hook)
// takes the 64-bit instantiations regardless: tests/test_raster_gpu.py runs the bit-exactness tests through both.
static bool hm_offsets_fit_32(int B, int S)
{
    static const bool force64 = [] { const char* e = getenv("HOMAN_FORCE_W64"); return e && atoi(e) != 0; }();
    return !force64 && 4.0 * B * (2.0 * S) * (2.0 * S) * sizeof(SweepSrc) < 4.0e9;
}
static void launch_lines(const SilWs& w, int B, int F, int S, int mode, const float* upstream, const float* keep_sum,
                         int clip_len, hipStream_t stream, float* loss_out = nullptr, int out_stride = 0)
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
static thread_local int g_sweep_blocks = SWEEP_BLOCKS;
static void launch_sweep(const SilWs& w, int B, int F, int S, float eps, int sum_log2q, hipStream_t stream)
{
    const int nsort = g_raster_reorder ? 1 : 0;      // (see hm_tune_raster_reorder: one workgroup more, eight workers less)
    const int blocks = max(8, (min(min(hm_cdiv((long)B * F, 2), g_sweep_blocks), TS_SWEEP_WGS - 8) & ~7) - 8 * nsort);   // workers: a multiple of 8, see the unit loop
    // (32-bit byte offsets while the largest array the sweep indexes - the per-line source arrays - stays below 4 GB)
    const bool w32 = hm_offsets_fit_32(B, S);
    hipLaunchKernelGGL(w32 ? k_bwd_sweep<true> : k_bwd_sweep<false>, dim3(nsort + blocks), dim3(256), 0, stream, w.sweep,
                       w.idx_map, w.srcs, w.lrec, B, F, S, eps, hm_sum_magic(sum_log2q), w.parts, w.lsum, w.alpha16, w.counter + 24,
                       w.ts + 2 * (ts_raster_units(B, S) + ts_lines_units(B, F, S)), nsort, w.wo_dyn, w.wo_tmp, w.wg_cost,
                       w.counter + 26, (int)ts_raster_units(B, S));
}

// Scheduling hint, no effect on results: bytes of unused dynamic LDS added to every k_raster_fwd launch.  The rasteriser's
// 24.9 KB of LDS and 80 registers fill a CU with 6 workgroups and leave nothing for the kernels of the caller's other
// stream (72 registers for the MANO forward: it then waits for the raster's tail); 3 KB of ballast caps the CU at 5
// workgroups.  Per calling thread (thread-local); read when hm_sil_fwd is called (or captured).  Returns the previous value; < 0 only queries.
static thread_local int g_raster_lds_pad = 0;
int hm_tune_raster_lds_pad(int bytes)
{
    const int prev = g_raster_lds_pad;
    if (bytes >= 0) g_raster_lds_pad = bytes;
    return prev;
}
// Scheduling hint, no effect on results: adaptive launch order of the forward raster (see g_raster_reorder).  enable > 0: on,
// 0: off, < 0: query.  Returns the previous value.  Per calling thread (thread-local); read when hm_sil_fwd / hm_sil_bwd are called (or captured).
int hm_tune_raster_reorder(int enable)
{
    const int prev = g_raster_reorder;
    if (enable >= 0) g_raster_reorder = enable ? 1 : 0;
    return prev;
}

// Scheduling hint, no effect on results: number of persistent workgroups of the edge-sweep kernel (default 1280 = 5 per
// CU).  The sweeps share the GPU with whatever runs on the caller's other streams; a loop whose other stream is the
// longer chain (collision + contact terms) finishes sooner with fewer sweep workgroups (768).  Per calling thread (thread-local); read when
// hm_sil_bwd is called (or captured).  Returns the previous value; blocks <= 0 only queries.
int hm_tune_sweep_blocks(int blocks)
{
    const int prev = g_sweep_blocks;
    if (blocks > 0) g_sweep_blocks = blocks;
    return prev;
}

// Forward: silhouettes (B,S,S) of `verts` under per-frame intrinsics K, optional fused masked-MSE/IoU.
//   keep/ref/keep_sum/loss_out may be NULL (render only).  loss_out[0]=loss_sil, loss_out[1]=mean IoU.
//   *_clips: the B frames are B / clip_len clips of clip_len frames (0: one clip); keep_sum and rigid_scale hold one
//   entry per clip, the loss / IoU of clip c go to loss_out[c * out_stride + 0 / 1].
//   *_phase_clips: the same forward in two calls for a caller that forks a second stream off the camera-space vertices:
//   phases = 1 launches the face setup only (cam_verts_out is complete when it ends), 2 the rasteriser (+ reduction) only,
//   3 both (= hm_sil_fwd_clips); both calls take the same arguments.
int hm_sil_fwd_phase_clips(const float* verts, const int* faces, int faces_bstride, const float* K, int B, int V, int F, int S,
                           float orig_size, float znear, float zfar, const float* keep, const float* ref,
                           const float* keep_sum, float* pooled, float* loss_out, const int* work_order, float* pooled_depth,
                           float* alpha_full, int mask_shared, const float* rigid_rot6d, const float* rigid_trans,
                           const float* rigid_scale, int rigid_abs, int persistent_outputs, void* workspace, int clip_len,
                           int out_stride, float* cam_verts_out, int phases, hipStream_t stream)
{
    HM_CHECK_ARG(phases >= 1 && phases <= 3);
    HM_CHECK_ARG(verts && faces && K && pooled && workspace && HM_CLIP_LEN_OK(B, clip_len));
    if (clip_len == 0) clip_len = B;
    HM_CHECK_ARG(!rigid_rot6d || (rigid_trans && rigid_scale));
    HM_CHECK_ARG(B > 0 && V > 0 && F > 0 && S > 0);
    if (S % 16 != 0 || 2 * S > 8192 || 2L * F >= (1L << 30) || B >= 32768) return HM_ERR_UNSUPPORTED;
    HM_CHECK_ARG(faces_bstride == 0 || faces_bstride == 3 * F);
    SilWs w = carve(workspace, B, V, F, S);
    const int is = 2 * S, ntiles = (S / 8) * (S / 8);
    int* bins = is <= (SR_MAX == 64 ? 1024 : 0) ? w.bin_cnt : nullptr;      // <= SR_MAX super-regions per frame
    HM_CHECK_ARG(!cam_verts_out || rigid_rot6d);
    const int nfb = hm_cdiv(F, 256);
    if (phases & 1)
        hipLaunchKernelGGL(k_setup_faces, dim3(nfb + (cam_verts_out ? hm_cdiv(V, 256) : 0), B), dim3(256), 0, stream, verts, K,
                           orig_size, faces, faces_bstride, B, V, F, is, w.faces9, w.boxes, w.owned, bins, w.bin_list,
                           rigid_rot6d, rigid_trans, rigid_scale, rigid_abs, clip_len, cam_verts_out, nfb);
    if (!(phases & 2)) return hm_launch_status();
    const bool fused = keep && ref;
    HM_TIME_MARK(0, stream);
    hipLaunchKernelGGL(k_raster_fwd, dim3(B * (ntiles / RASTER_WAVES)), dim3(64 * RASTER_WAVES), g_raster_lds_pad, stream,
                       w.faces9, w.boxes, B, F, S, znear, zfar, w.idx_map, w.alpha16, pooled, keep, ref, w.dimg,
                       fused ? w.partials : (float*)nullptr, work_order, w.owned, pooled_depth, w.planes, bins,
                       w.bin_list, w.bin_done, 1, w.region_state, persistent_outputs, alpha_full, mask_shared,
                       (fused && (alpha_full || (mask_shared & 2))) ? w.gimg : (float*)nullptr, w.counter + 24, w.ts,
                       g_raster_reorder ? w.wo_dyn : (int*)nullptr, g_raster_reorder ? w.wg_cost : (unsigned int*)nullptr);
    HM_TIME_MARK(1, stream);
    if (fused && keep_sum && loss_out)
        hipLaunchKernelGGL(k_sil_reduce, dim3(B), dim3(256), 0, stream, w.partials, B, ntiles, keep_sum, w.frame_rec,
                           loss_out, (float*)nullptr, clip_len, out_stride);
    return hm_launch_status();
}
int hm_sil_fwd_clips(const float* verts, const int* faces, int faces_bstride, const float* K, int B, int V, int F, int S,
                     float orig_size, float znear, float zfar, const float* keep, const float* ref,
                     const float* keep_sum, float* pooled, float* loss_out, const int* work_order, float* pooled_depth,
                     float* alpha_full, int mask_shared, const float* rigid_rot6d, const float* rigid_trans,
                     const float* rigid_scale, int rigid_abs, int persistent_outputs, void* workspace, int clip_len,
                     int out_stride, float* cam_verts_out, hipStream_t stream)
{
    return hm_sil_fwd_phase_clips(verts, faces, faces_bstride, K, B, V, F, S, orig_size, znear, zfar, keep, ref, keep_sum, pooled,
                                  loss_out, work_order, pooled_depth, alpha_full, mask_shared, rigid_rot6d, rigid_trans,
                                  rigid_scale, rigid_abs, persistent_outputs, workspace, clip_len, out_stride, cam_verts_out, 3,
                                  stream);
}
int hm_sil_fwd(const float* verts, const int* faces, int faces_bstride, const float* K, int B, int V, int F, int S,
               float orig_size, float znear, float zfar, const float* keep, const float* ref,
               const float* keep_sum, float* pooled, float* loss_out, const int* work_order, float* pooled_depth,
               float* alpha_full, int mask_shared, const float* rigid_rot6d, const float* rigid_trans,
               const float* rigid_scale, int rigid_abs, int persistent_outputs, void* workspace, hipStream_t stream)
{
    return hm_sil_fwd_clips(verts, faces, faces_bstride, K, B, V, F, S, orig_size, znear, zfar, keep, ref, keep_sum, pooled,
                            loss_out, work_order, pooled_depth, alpha_full, mask_shared, rigid_rot6d, rigid_trans,
                            rigid_scale, rigid_abs, persistent_outputs, workspace, 0, 0, nullptr, stream);
}

// Scheduling hint, no effect on results: which winding class of the mesh (0: faces as stored, 1: reversed copies of
// fill_back) holds the camera-facing surface.  The forward rasterises that class first and tests the units of the other
// class against per-block hidden depths before doing any per-sample work: on a closed mesh the far class owns nothing.
// Stored in the workspace (stream-ordered 4-byte write); the zero-filled default is class 0.
int hm_sil_hint_near_winding(void* workspace, int winding, hipStream_t stream)
{
    HM_CHECK_ARG(workspace && (winding == 0 || winding == 1));
    return hipMemsetAsync((char*)workspace + 24 * 4, winding, 1, stream) == hipSuccess ? HM_OK : HM_ERR_LAUNCH;
}

// The loss / IoU reduction of a forward that was called with keep/ref but loss_out == NULL: the backward does not
// depend on it, so a caller with a second stream takes it off the critical path.
// frame_out (B,2) optional: per-frame {sum of squares (un-normalised), IoU}; loss_out may then be NULL.
int hm_sil_reduce_clips(int B, int V, int F, int S, const float* keep_sum, float* loss_out, float* frame_out,
                        void* workspace, int clip_len, int out_stride, hipStream_t stream)
{
    HM_CHECK_ARG(workspace && B > 0 && S > 0 && (frame_out || loss_out) && (!loss_out || keep_sum));
    HM_CHECK_ARG(HM_CLIP_LEN_OK(B, clip_len));
    SilWs w = carve(workspace, B, V, F, S);
    hipLaunchKernelGGL(k_sil_reduce, dim3(B), dim3(256), 0, stream, w.partials, B, (S / 8) * (S / 8), keep_sum,
                       w.frame_rec, loss_out, frame_out, clip_len ? clip_len : B, out_stride);
    return hm_launch_status();
}
int hm_sil_reduce(int B, int V, int F, int S, const float* keep_sum, float* loss_out, float* frame_out, void* workspace,
                  hipStream_t stream)
{
    return hm_sil_reduce_clips(B, V, F, S, keep_sum, loss_out, frame_out, workspace, 0, 0, stream);
}

// Backward.  mode 1 (fused loss): upstream = d/d loss_sil (device scalar), uses dimg from the forward.
//            mode 2: as mode 1, and the caller guarantees upstream[0] > 0 (one launch less).
//            mode 0 (render):     grad_pooled (B,S,S) = dL/d silhouettes.
//            mode 3 (render without anti-aliasing): grad_pooled is (B,2S,2S) = dL/d alpha_full.
//            mode 4 (fused per-sample L2 of a forward called with alpha_full + keep/ref): upstream (B) = dL/d frame sums, all > 0.
// adjacency (CSR over V) describes the shared face topology.  grad_verts (B,V,3) is overwritten.
//   *_clips (modes 1 / 2): keep_sum holds one entry per clip of clip_len frames and the 1/B of the loss is 1/clip_len;
//   `upstream` stays one scalar shared by the clips.  loss_out (optional, modes 1 / 2): the loss / IoU reduction of a forward
//   called with keep / ref but loss_out == NULL (what hm_sil_reduce_clips computes, same arithmetic) rides at the front of
//   the backward's first launch: clip c's values at loss_out[c * out_stride + 0 / 1].
// phases: bit 0 = sample-gradient masks (generic modes) + line expansion + work list, bit 1 = edge sweeps + vertex gather: the
// backward in two calls for a caller that lets other streams wait for the END of the line expansion (the kernel of the chain
// that suffers most from latency-bound neighbours holding its wave slots); hm_sil_bwd_clips = both
int hm_sil_bwd_phase_clips(const float* verts, const float* K, int B, int V, int F, int S, float orig_size, float eps, int mode,
                           const float* upstream, const float* grad_pooled, const float* keep_sum, const int* adj_off,
                           const int* adj_items, const int* face_order, float* grad_verts, float* grad_ndc, void* workspace,
                           int clip_len, float* loss_out, int out_stride, int phases, int sum_log2q, hipStream_t stream)
{
    HM_CHECK_ARG(!loss_out || ((mode == 1 || mode == 2) && keep_sum));
    HM_CHECK_ARG(sum_log2q <= 0 && sum_log2q >= -60);
    HM_CHECK_ARG(verts && K && adj_off && adj_items && workspace);        // grad_verts == NULL: no vertex gather (see hm_sil_parts)
    HM_CHECK_ARG(HM_CLIP_LEN_OK(B, clip_len));
    if (clip_len == 0) clip_len = B;
    HM_CHECK_ARG((mode == 0 || mode == 3) ? grad_pooled != nullptr : ((mode == 4 || mode == 5) ? upstream != nullptr : (upstream && keep_sum)));
    HM_CHECK_ARG(mode >= 0 && mode <= 5);
    if (S % 32 != 0 || S > 32 * SWEEP_CUMW) return HM_ERR_UNSUPPORTED;     // 64-sample mask words, <= SWEEP_CUMW per line
    SilWs w = carve(workspace, B, V, F, S);
    const int ntiles = (S / 8) * (S / 8);
    HM_CHECK_ARG(phases >= 1 && phases <= 3);
    if ((phases & 1) && mode != 2 && mode != 4 && mode != 5)      // modes 2 / 4 / 5: the caller guarantees upstream > 0, the forward's planes are the backward's
        hipLaunchKernelGGL(k_bwd_masks, dim3(hm_cdiv(ntiles, 4), B), dim3(256), 0, stream,
                           mode == 1 ? w.dimg : grad_pooled, mode, upstream, keep_sum, B, S, w.alpha16, w.gimg,
                           w.planes, clip_len);
    HM_TIME_MARK(2, stream);
    if (phases & 1) launch_lines(w, B, F, S, mode, upstream, keep_sum, clip_len, stream, loss_out, out_stride);
    HM_TIME_MARK(3, stream);
    if (!(phases & 2)) return hm_launch_status();
    launch_sweep(w, B, F, S, eps, sum_log2q, stream);
    HM_TIME_MARK(4, stream);
    if (grad_verts)
        hipLaunchKernelGGL(k_bwd_gather, dim3(hm_cdiv((long)B * V, 256)), dim3(256), 0, stream, w.parts, adj_off,
                           adj_items, verts, K, B, V, F, orig_size, grad_ndc, grad_verts);
    return hm_launch_status();
}
int hm_sil_bwd_clips(const float* verts, const float* K, int B, int V, int F, int S, float orig_size, float eps, int mode,
                     const float* upstream, const float* grad_pooled, const float* keep_sum, const int* adj_off,
                     const int* adj_items, const int* face_order, float* grad_verts, float* grad_ndc, void* workspace,
                     int clip_len, float* loss_out, int out_stride, int sum_log2q, hipStream_t stream)
{
    return hm_sil_bwd_phase_clips(verts, K, B, V, F, S, orig_size, eps, mode, upstream, grad_pooled, keep_sum, adj_off, adj_items,
                                  face_order, grad_verts, grad_ndc, workspace, clip_len, loss_out, out_stride, 3, sum_log2q, stream);
}
int hm_sil_bwd(const float* verts, const float* K, int B, int V, int F, int S, float orig_size, float eps, int mode,
               const float* upstream, const float* grad_pooled, const float* keep_sum, const int* adj_off,
               const int* adj_items, const int* face_order, float* grad_verts, float* grad_ndc, void* workspace,
               int sum_log2q, hipStream_t stream)
{
    return hm_sil_bwd_clips(verts, K, B, V, F, S, orig_size, eps, mode, upstream, grad_pooled, keep_sum, adj_off, adj_items,
                            face_order, grad_verts, grad_ndc, workspace, 0, nullptr, 0, sum_log2q, stream);
}

// (B,F,3,2) DOUBLES: d loss / d NDC (x, y) per face corner, as left by the last hm_sil_bwd (exact sums of terms on the
// grid 2^sum_log2q, see hm_quant): input of hm_rigid_bwd_sil.
const double* hm_sil_parts(const void* workspace, int B, int V, int F, int S)
{
    return carve((void*)workspace, B, V, F, S).parts;
}

// rgb image (B,3,S,S) of the last hm_sil_fwd on this workspace (same verts / faces): per-face colours `textures`
// (B,F,3) under flat lighting, as nr.renderer.Renderer.render returns it for texture_size 1 (reference
// homan/homan.py:535-538 with the light of :173-176).  light_dir / background: HOST float[3].
int hm_shade_rgb(const float* verts, const int* faces, int faces_bstride, const float* textures, int B, int V, int F, int S,
                 const float* light_dir, float intensity_ambient, float intensity_directional, const float* background,
                 float* rgb, void* workspace, hipStream_t stream)
{
    HM_CHECK_ARG(verts && faces && textures && light_dir && background && rgb && workspace);
    HM_CHECK_ARG(B > 0 && V > 0 && F > 0 && S > 0);
    SilWs w = carve(workspace, B, V, F, S);
    hipLaunchKernelGGL(k_shade_rgb, dim3(hm_cdiv((long)S * S, 256), B), dim3(256), 0, stream, w.idx_map, verts, faces,
                       faces_bstride, textures, B, V, F, S, light_dir[0], light_dir[1], light_dir[2], intensity_ambient,
                       intensity_directional, background[0], background[1], background[2], rgb);
    return hm_launch_status();
}

// Backward of the depth image of the last hm_sil_fwd (called with pooled_depth): grad_pooled_depth (B,S,S) ->
// grad_verts (B,V,3).  Uses faces9 / boxes / idx_map / owned of the workspace; gf9 scratch lives in `parts`.
int hm_depth_bwd(const float* verts, const float* K, int B, int V, int F, int S, float orig_size,
                 const float* grad_pooled_depth, const int* adj_off, const int* adj_items, float* grad_verts,
                 void* workspace, hipStream_t stream)
{
    HM_CHECK_ARG(verts && K && grad_pooled_depth && adj_off && adj_items && grad_verts && workspace);
    if (S % 16 != 0) return HM_ERR_UNSUPPORTED;
    SilWs w = carve(workspace, B, V, F, S);
    hipLaunchKernelGGL(k_depth_bwd_faces, dim3(hm_cdiv(hm_cdiv((long)B * F, DBF_FACES) * 64, 256)), dim3(256), 0, stream, w.faces9, w.boxes,
                       w.idx_map, grad_pooled_depth, w.owned, B, F, S, (float*)w.parts);
    hipLaunchKernelGGL(k_depth_bwd_gather, dim3(hm_cdiv((long)B * V, 256)), dim3(256), 0, stream, (const float*)w.parts, adj_off,
                       adj_items, verts, K, B, V, F, orig_size, grad_verts);
    return hm_launch_status();
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
int hm_ordinal_depth_bwd(const float* d0, const float* d1, const float* a0, const float* a1, const unsigned char* m0,
                         const unsigned char* m1, int B, int S, const float* rec, const float* upstream, float* g0,
                         float* g1, hipStream_t stream)
{
    HM_CHECK_ARG(d0 && d1 && a0 && a1 && m0 && m1 && rec && upstream && g0 && g1);
    const long n = (long)B * S * S;
    hipLaunchKernelGGL(k_ordinal_depth_bwd, dim3(hm_cdiv(n, 256)), dim3(256), 0, stream, d0, d1, a0, a1, m0, m1, n, rec,
                       upstream, g0, g1);
    return hm_launch_status();
}

// Measurement hook for bench.py: runs one full forward + backward (fused-loss mode, upstream = 1) to populate the
// workspace, then `reps` launches of k_raster_fwd alone and `reps` launches of k_bwd_sweep alone, each bracketed by two
// HIP events recorded on `stream`; avg_ms[0] / avg_ms[1] (HOST pointer) receive the average launch durations in
// milliseconds.  Synchronises.
int hm_bench_sil_kernels(const float* verts, const int* faces, const float* K, int B, int V, int F, int S,
                         const float* keep, const float* ref, const float* keep_sum, float* pooled, float* loss_out,
                         const int* work_order, const int* adj_off, const int* adj_items, const int* face_order,
                         const float* upstream, float* grad_verts, void* workspace, int reps, float* avg_ms,
                         hipStream_t stream)
{
    HM_CHECK_ARG(verts && faces && K && keep && ref && keep_sum && pooled && loss_out && workspace && reps > 0 && avg_ms);
    HM_CHECK_ARG(adj_off && adj_items && upstream && grad_verts);
    int rc = hm_sil_fwd(verts, faces, 0, K, B, V, F, S, 1.0f, 0.1f, 100.0f, keep, ref, keep_sum, pooled, loss_out,
                        work_order, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, workspace, stream);
    if (rc != HM_OK) return rc;
    rc = hm_sil_bwd(verts, K, B, V, F, S, 1.0f, 1e-3f, 1, upstream, nullptr, keep_sum, adj_off, adj_items, face_order,
                    grad_verts, nullptr, workspace, 0, stream);
    if (rc != HM_OK) return rc;
    SilWs w = carve(workspace, B, V, F, S);
    const int ntiles = (S / 8) * (S / 8);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return HM_ERR_LAUNCH;
    float ms = 0.f;
    // the forward's last workgroup emptied the super-region bins: fill them again and keep them across the timed launches
    int* bins = 2 * S <= 1024 ? w.bin_cnt : nullptr;
    hipLaunchKernelGGL(k_setup_faces, dim3(hm_cdiv(F, 256), B), dim3(256), 0, stream, verts, K, 1.0f, faces, 0, B, V, F,
                       2 * S, w.faces9, w.boxes, w.owned, bins, w.bin_list, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, 0, B, (float*)nullptr, hm_cdiv(F, 256));
    const bool cold = false;   // true: re-bin before every launch (times setup + raster with the reset tickets)
    (void)hipEventRecord(e0, stream);
    for (int i = 0; i < reps; ++i) {
        if (cold) {
            (void)hipMemsetAsync(w.bin_cnt, 0, (size_t)B * SR_MAX * 4, stream);
            hipLaunchKernelGGL(k_setup_faces, dim3(hm_cdiv(F, 256), B), dim3(256), 0, stream, verts, K, 1.0f, faces, 0, B, V, F,
                               2 * S, w.faces9, w.boxes, w.owned, bins, w.bin_list, (const float*)nullptr,
                               (const float*)nullptr, (const float*)nullptr, 0, B, (float*)nullptr, hm_cdiv(F, 256));
        }
        hipLaunchKernelGGL(k_raster_fwd, dim3(B * (ntiles / RASTER_WAVES)), dim3(64 * RASTER_WAVES), 0, stream,
                           w.faces9, w.boxes, B, F, S, 0.1f, 100.0f, w.idx_map, w.alpha16, pooled, keep, ref, w.dimg,
                           w.partials, work_order, w.owned, (float*)nullptr, w.planes, bins, w.bin_list,
                           w.bin_done, cold ? 1 : 0, w.region_state, 1, (float*)nullptr, 0, (float*)nullptr,
                           w.counter + 24, w.ts, (int*)nullptr, (unsigned int*)nullptr);      // steady state of a fixed loop: background regions skipped
    }
    (void)hipEventRecord(e1, stream);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    avg_ms[0] = ms / (float)reps;
    (void)hipMemsetAsync(w.bin_cnt, 0, (size_t)B * SR_MAX * 4, stream);
    (void)hipEventRecord(e0, stream);
    for (int i = 0; i < reps; ++i) launch_sweep(w, B, F, S, 1e-3f, 0, stream);
    (void)hipEventRecord(e1, stream);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    avg_ms[1] = ms / (float)reps;
    (void)hipEventRecord(e0, stream);
    for (int i = 0; i < reps; ++i) launch_lines(w, B, F, S, 1, upstream, keep_sum, B, stream);
    (void)hipEventRecord(e1, stream);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    avg_ms[2] = ms / (float)reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return hm_launch_status();
}

// enable != 0: from now on hm_sil_fwd / hm_sil_bwd record HIP events around k_raster_fwd, k_bwd_lines and k_bwd_sweep on
// their launch stream (do not enable while a stream capture is in progress); 0: stop and release the events.
int hm_debug_sil_timing(int enable)
{
    if (enable && !g_timing) {
        for (int k = 0; k < 5; ++k)
            if (hipEventCreate(&g_tev[k]) != hipSuccess) return HM_ERR_LAUNCH;
        g_timing = 1;
    } else if (!enable && g_timing) {
        g_timing = 0;
        for (int k = 0; k < 5; ++k) (void)hipEventDestroy(g_tev[k]);
    }
    return HM_OK;
}
// In-graph timing of the same three kernels: device wall-clock stamps stored by the workgroups themselves (see
// hm_ts_enabled above), so the numbers come from launches replayed from a captured hipGraph - where ROCm allows no events.
//   hm_sil_timestamps(ws, B, V, F, S, 1, stream): switch on and arm (two async memsets, no host synchronisation); call
//   before every replay;  (..., 0, stream): switch off
//   hm_sil_timestamps_bytes(B, V, F, S): size of one record of raw stamps
//   hm_sil_timestamps_save(ws, B, V, F, S, dst, stream): async device copy of the raw stamps to `dst` - a caller that must
//   not synchronise between replays saves every iteration's record and reads them all at the end
//   hm_sil_timestamps_read(ws, B, V, F, S, saved, us3, stream): waits for the stream; durations (earliest start to latest
//   end over the workgroups) of k_raster_fwd, k_bwd_lines, k_bwd_sweep in microseconds -> us3 (HOST), from `saved` (a
//   hm_sil_timestamps_save record) or, saved == NULL, from the workspace itself (0 for a kernel that did not run)
size_t hm_sil_timestamps_bytes(int B, int V, int F, int S) { (void)V; return ts_units(B, F, S) * 16; }
int hm_sil_timestamps(void* workspace, int B, int V, int F, int S, int enable, hipStream_t stream)
{
    HM_CHECK_ARG(workspace && B > 0 && F > 0 && S > 0);
    SilWs w = carve(workspace, B, V, F, S);
    if (hipMemsetAsync((char*)workspace + 24 * 4 + 1, enable ? 1 : 0, 1, stream) != hipSuccess) return HM_ERR_LAUNCH;
    if (enable && hipMemsetAsync(w.ts, 0, ts_units(B, F, S) * 16, stream) != hipSuccess) return HM_ERR_LAUNCH;
    return HM_OK;
}
int hm_sil_timestamps_save(const void* workspace, int B, int V, int F, int S, void* dst, hipStream_t stream)
{
    HM_CHECK_ARG(workspace && dst);
    SilWs w = carve((void*)workspace, B, V, F, S);
    return hipMemcpyAsync(dst, w.ts, ts_units(B, F, S) * 16, hipMemcpyDeviceToDevice, stream) == hipSuccess ? HM_OK : HM_ERR_LAUNCH;
}
int hm_sil_timestamps_read(const void* workspace, int B, int V, int F, int S, const void* saved, float* us3, hipStream_t stream)
{
    HM_CHECK_ARG((workspace || saved) && us3 && B > 0 && F > 0 && S > 0);
    const size_t n = ts_units(B, F, S);
    unsigned long long* t = (unsigned long long*)malloc(n * 16);
    if (!t) return HM_ERR_LAUNCH;
    const void* src = saved ? saved : (const void*)carve((void*)workspace, B, V, F, S).ts;
    if (hipMemcpyAsync(t, src, n * 16, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
        free(t);
        return HM_ERR_LAUNCH;
    }
    int dev = 0, khz = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    const size_t lo[4] = {0, ts_raster_units(B, S), ts_raster_units(B, S) + ts_lines_units(B, F, S), n};
    for (int k = 0; k < 3; ++k) {
        unsigned long long t0 = ~0ull, t1 = 0ull;
        for (size_t u = lo[k]; u < lo[k + 1]; ++u) {
            if (t[2 * u] != 0ull && t[2 * u] < t0) t0 = t[2 * u];
            if (t[2 * u + 1] > t1) t1 = t[2 * u + 1];
        }
        us3[k] = (t0 != ~0ull && t1 > t0) ? (float)((double)(t1 - t0) * 1e3 / (double)khz) : 0.f;
    }
    free(t);
    return HM_OK;
}
// durations (ms) of the LAST timed k_raster_fwd, k_bwd_lines, k_bwd_sweep launches -> ms3 (HOST pointer).  Waits for them.
int hm_debug_sil_timing_read(float* ms3)
{
    HM_CHECK_ARG(ms3 && g_timing);
    if (hipEventSynchronize(g_tev[4]) != hipSuccess) return HM_ERR_LAUNCH;
    if (hipEventElapsedTime(ms3, g_tev[0], g_tev[1]) != hipSuccess) return HM_ERR_LAUNCH;
    if (hipEventElapsedTime(ms3 + 1, g_tev[2], g_tev[3]) != hipSuccess) return HM_ERR_LAUNCH;
    if (hipEventElapsedTime(ms3 + 2, g_tev[3], g_tev[4]) != hipSuccess) return HM_ERR_LAUNCH;
    return HM_OK;
}
int hm_debug_read_partials(const void* workspace, int B, int V, int F, int S, float* out, hipStream_t stream)
{
    SilWs w = carve((void*)workspace, B, V, F, S);
    return hipMemcpyAsync(out, w.partials, (size_t)B * (S / 8) * (S / 8) * 16, hipMemcpyDeviceToDevice, stream) == hipSuccess
               ? HM_OK : HM_ERR_LAUNCH;
}
#ifdef RASTER_PHASES
int hm_debug_raster_phases(unsigned long long* out)
{
    unsigned long long z[24] = {0};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_raster_ph), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_raster_ph), z, sizeof(z));
    return HM_OK;
}
#endif
#ifdef SWEEP_UNIT_PROFILE
int hm_debug_unit_profile(int* out)              // 65536 x 4 ints, then cleared
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_unit_prof), sizeof(int) * 65536 * 4);
    static int z[65536][4];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_unit_prof), z, sizeof(z));
    return HM_OK;
}
#endif
#ifdef SWEEP_STATS
int hm_debug_sweep_stats(unsigned long long* out)
{
    unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sweep_n), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sweep_n), z, sizeof(z));
    return HM_OK;
}
#endif
// test hook: cap > 0 shrinks the unit table and the partial-slot table of the sweep work list to `cap` entries (the
// workspace keeps its size), so that small inputs exercise the beyond-capacity paths; 0 restores the defaults.
int hm_debug_sweep_caps(int cap)
{
    const int prev = g_sweep_cap_override;
    g_sweep_cap_override = cap > 0 ? cap : 0;
    return prev;
}
int hm_debug_occupancy(int* raster_fwd_blocks, int* sweep_blocks)
{
    hipError_t e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(raster_fwd_blocks, k_raster_fwd, 64 * RASTER_WAVES, 0);
    hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(sweep_blocks, k_bwd_sweep<true>, 256, 0);
    return (e1 == hipSuccess && e2 == hipSuccess) ? HM_OK : HM_ERR_LAUNCH;
}

// debug / test access to forward intermediates held in the workspace
int hm_sil_read_idx_map(const void* workspace, int B, int V, int F, int S, int* out, hipStream_t stream)
{
    SilWs w = carve((void*)workspace, B, V, F, S);
    return hipMemcpyAsync(out, w.idx_map, (size_t)B * 4 * S * S * 4, hipMemcpyDeviceToDevice, stream) == hipSuccess
               ? HM_OK : HM_ERR_LAUNCH;
}
int hm_sil_read_boxes(const void* workspace, int B, int V, int F, int S, void* out, hipStream_t stream)
{
    SilWs w = carve((void*)workspace, B, V, F, S);
    return hipMemcpyAsync(out, w.boxes, (size_t)B * F * 8, hipMemcpyDeviceToDevice, stream) == hipSuccess ? HM_OK
                                                                                                         : HM_ERR_LAUNCH;
}
int hm_sil_read_faces9(const void* workspace, int B, int V, int F, int S, float* out, hipStream_t stream)
{
    SilWs w = carve((void*)workspace, B, V, F, S);
    return hipMemcpyAsync(out, w.faces9, (size_t)B * F * 9 * 4, hipMemcpyDeviceToDevice, stream) == hipSuccess
               ? HM_OK : HM_ERR_LAUNCH;
}
// hm_sil_fwd with persistent_outputs skips the epilogue of an empty region whose outputs already hold the empty pattern - which
// depends on the loss inputs (keep / ref).  A caller that REUSES a workspace and its output buffers for another clip (new
// masks in the same buffers) calls this once after loading them: every region writes its outputs again on the next forward.
int hm_sil_invalidate_outputs(void* workspace, int B, int V, int F, int S, hipStream_t stream)
{
    HM_CHECK_ARG(workspace && B > 0 && S > 0);
    SilWs w = carve(workspace, B, V, F, S);
    return hipMemsetAsync(w.region_state, 0, (size_t)B * (S / 16) * (S / 16), stream) == hipSuccess ? HM_OK : HM_ERR_LAUNCH;
}
// (B,F,3,2) doubles: the per-(face, corner) sums of the last backward (tests: compared bit for bit with the CPU oracle)
int hm_sil_read_parts(const void* workspace, int B, int V, int F, int S, double* out, hipStream_t stream)
{
    SilWs w = carve((void*)workspace, B, V, F, S);
    return hipMemcpyAsync(out, w.parts, (size_t)B * F * 6 * 8, hipMemcpyDeviceToDevice, stream) == hipSuccess
               ? HM_OK : HM_ERR_LAUNCH;
}
#ifdef HM_CHAIN_STAMPS
int hm_debug_chain_raster(unsigned long long* out, int reset)
{
    unsigned long long z[8] = {~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0};
    (void)hipDeviceSynchronize();
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_ts), sizeof(z));
    if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chain_ts), z, sizeof(z));
    return HM_OK;
}
#endif
}  // extern "C"
