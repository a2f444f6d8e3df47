// adam.hip -- fused multi-tensor Adam step and sync-free scalar logging for the optimisation loop.
//
// Counterpart of the `torch.optim.Adam([...3 groups...])` step at reference homan/jointopt.py:138-151,192, restating
// torch's single-tensor Adam arithmetic (the implementation the reference's CPU path runs) in fp32:
//   m = m + (1-b1) (g - m);  v = v*b2 + (1-b2) g g;  step_size = lr / (1 - b1^t);
//   denom = sqrt(v) / sqrt(1 - b2^t) + eps;  p = p + (-step_size * m) / denom
// with the bias corrections evaluated in double like the Python scalars they are in torch.  One launch updates every
// parameter tensor (pointer table in device memory), zeroes the gradients for the next iteration, and the step
// counter lives on the device so the whole iteration can sit in a hipGraph.
#include "hm_common.h"

struct AdamSlot {
    float* p; float* g; float* m; float* v;
    long n;
    float lr;
    int pad;
};

// grid (blocks_per_tensor, n_tensors)
// step[0] = completed steps, step[1] = ticket word (zero between launches): the workgroup that draws the last ticket
// bumps the step counter, after every workgroup has read it -- no separate increment launch on the critical path.
// Optional log row (vals != NULL): one more grid row (blockIdx.y == n_tensors) writes what k_log_total writes - the weighted
// total of every clip and row `step` of the log - before it draws its tickets: the step counter moves only after every
// workgroup has read it, so the row lands in the slot of the step being taken, and the iteration has one launch less.
__device__ __forceinline__ void log_total_clip(float* __restrict__ vals, const float* __restrict__ w, int n, int s,
                                               int max_steps, float* __restrict__ log, int c, int nc)
{
    float* v = vals + (long)c * (n + 1);
    float t = 0.f;
    for (int i = 0; i < n; ++i)
        if (w[i] != 0.f) t += w[i] * v[i];
    v[n] = t;
    if (s < max_steps)
        for (int i = 0; i <= n; ++i) log[((long)s * nc + c) * (n + 1) + i] = v[i];
}
__device__ __forceinline__ double adam_pow(double b, int t)
{
    double r = 1.0;
    for (; t > 0; t >>= 1) {
        if (t & 1) r = r * b;
        b = b * b;
    }
    return r;
}
__global__ __launch_bounds__(256) void k_adam(const AdamSlot* __restrict__ slots, int n_tensors, int* __restrict__ step,
                                               float beta1, float beta2, float eps, int zero_grad, float* __restrict__ vals,
                                               const float* __restrict__ log_w, int log_n, int max_steps,
                                               float* __restrict__ log, int nclips)
{
    HM_LATENCY_KERNEL();
    HM_STAMP_START(0);
    const int step_now = __builtin_nontemporal_load(step);
    if ((int)blockIdx.y == n_tensors) {
        if (threadIdx.x == 0) {
            for (int c = blockIdx.x; c < nclips; c += gridDim.x) log_total_clip(vals, log_w, log_n, step_now, max_steps, log, c, nclips);
            const unsigned nblk = gridDim.x * gridDim.y;
            if (atomicAdd(reinterpret_cast<unsigned int*>(step) + 1, 1u) == nblk - 1u) {
                atomicExch(reinterpret_cast<unsigned int*>(step) + 1, 0u);
                atomicExch(step, step_now + 1);
            }
        }
        return;
    }
    const AdamSlot s = slots[blockIdx.y];
    // beta^t by square-and-multiply in double: IEEE products in a fixed order, i.e. a function of (beta, t) alone - the CPU
    // oracle's Adam (oracle/adam.py) forms the same doubles, where a libm pow() may differ in the last bit from host to host
    const double bc1 = 1.0 - adam_pow((double)beta1, step_now + 1);
    const double bc2 = 1.0 - adam_pow((double)beta2, step_now + 1);
    const float neg_step = (float)(-((double)s.lr / bc1));
    const float bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - (double)beta1), w2 = (float)(1.0 - (double)beta2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < s.n; i += (long)gridDim.x * blockDim.x) {
        const float g = s.g[i];
        float m = s.m[i], v = s.v[i];
        m = m + w1 * (g - m);
        v = v * beta2;
        v = v + (w2 * g) * g;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        s.p[i] = s.p[i] + (neg_step * m) / denom;
        s.m[i] = m;
        s.v[i] = v;
        if (zero_grad) s.g[i] = 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned nblk = gridDim.x * gridDim.y;
        if (atomicAdd(reinterpret_cast<unsigned int*>(step) + 1, 1u) == nblk - 1u) {
            atomicExch(reinterpret_cast<unsigned int*>(step) + 1, 0u);
            atomicExch(step, step_now + 1);
        }
    }
    HM_STAMP_END(0);
}

// log[step*n + i] = src[i] for i < n ; step read from the device counter (sync-free loss_evolution)
__global__ void k_log(const float* __restrict__ src, int n, const int* __restrict__ step, int max_steps,
                      float* __restrict__ log)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = step[0];
    if (i < n && t < max_steps) log[(long)t * n + i] = src[i];
}

// per clip c (grid (clips)): total = sum_i w[i]*vals[c][i] (i < n) -> vals[c][n] ; then row (step, c) of the log =
// vals[c][0..n] (n+1 floats).  vals (clips, n+1), log (max_steps, clips, n+1).
__global__ void k_log_total(float* __restrict__ vals, const float* __restrict__ w, int n, const int* __restrict__ step,
                            int max_steps, float* __restrict__ log)
{
    HM_LATENCY_KERNEL();
    if (threadIdx.x == 0) log_total_clip(vals, w, n, step[0], max_steps, log, blockIdx.x, gridDim.x);
}

extern "C" {
int hm_log_total_clips(float* vals, const float* weights, int n, const int* step, int max_steps, float* log, int nclips,
                       hipStream_t stream)
{
    HM_CHECK_ARG(vals && weights && step && log && n > 0 && nclips > 0);
    hipLaunchKernelGGL(k_log_total, dim3(nclips), dim3(64), 0, stream, vals, weights, n, step, max_steps, log);
    return hm_launch_status();
}
int hm_log_total(float* vals, const float* weights, int n, const int* step, int max_steps, float* log,
                 hipStream_t stream)
{
    return hm_log_total_clips(vals, weights, n, step, max_steps, log, 1, stream);
}
size_t hm_adam_slot_bytes(void) { return sizeof(AdamSlot); }

int hm_adam_step(const void* slots, int n_tensors, int* step, float beta1, float beta2, float eps, int zero_grad,
                 int blocks_per_tensor, hipStream_t stream)
{
    HM_CHECK_ARG(slots && step && n_tensors > 0 && blocks_per_tensor > 0);
    hipLaunchKernelGGL(k_adam, dim3(blocks_per_tensor, n_tensors), dim3(256), 0, stream, (const AdamSlot*)slots, n_tensors,
                       step, beta1, beta2, eps, zero_grad, (float*)nullptr, (const float*)nullptr, 0, 0, (float*)nullptr, 0);
    return hm_launch_status();
}
// hm_log_total_clips + hm_adam_step in ONE launch (same values, same log row: the row is written before the step counter
// moves): the log of the step being taken costs no launch of its own.
int hm_adam_step_log(const void* slots, int n_tensors, int* step, float beta1, float beta2, float eps, int zero_grad,
                     int blocks_per_tensor, float* vals, const float* weights, int n, int max_steps, float* log, int nclips,
                     hipStream_t stream)
{
    HM_CHECK_ARG(slots && step && n_tensors > 0 && blocks_per_tensor > 0);
    HM_CHECK_ARG(vals && weights && log && n > 0 && nclips > 0);
    hipLaunchKernelGGL(k_adam, dim3(blocks_per_tensor, n_tensors + 1), dim3(256), 0, stream, (const AdamSlot*)slots, n_tensors,
                       step, beta1, beta2, eps, zero_grad, vals, weights, n, max_steps, log, nclips);
    return hm_launch_status();
}
int hm_log_scalars(const float* src, int n, const int* step, int max_steps, float* log, hipStream_t stream)
{
    HM_CHECK_ARG(src && step && log && n > 0);
    hipLaunchKernelGGL(k_log, dim3(hm_cdiv(n, 64)), dim3(64), 0, stream, src, n, step, max_steps, log);
    return hm_launch_status();
}
#ifdef HM_CHAIN_STAMPS
int hm_debug_chain_adam(unsigned long long* out, int reset)
{
    unsigned long long z[8] = {~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0};
    (void)hipDeviceSynchronize();
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_ts), sizeof(z));
    if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chain_ts), z, sizeof(z));
    return HM_OK;
}
#endif
}  // extern "C"
