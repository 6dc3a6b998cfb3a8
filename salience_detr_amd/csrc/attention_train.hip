// Dense multi-head attention over a few hundred rows for the TRAINING step, forward and backward (fp32, 32-channel heads):
// the encoder layer's self-attention over its top-300 rows (models/bricks/salience_transformer.py:371-376,
// nn.MultiheadAttention(256, 8) with q = k = x + pos, v = x) between its in- and out-projection.
//
// Under autograd the framework spends ~1 ms of the 18 ms step on these six 300-row problems: three batched products each
// way, softmax and its backward, and ~20 layout copies / zero fills per layer around the head split -- all launch-bound.
// Here the projected rows are read where the two in-projection GEMMs leave them (q | k of a row side by side, heads along
// the features) and the heads come out concatenated for the out-projection: no copy, three launches per layer in all.
//   forward      : workgroup = (32 query rows, head, image), 8 lanes per row (160 workgroups for 2 x 300 rows: with 4 lanes
//                  per row the 80 single-wave-per-SIMD workgroups were instruction-bound at 34 us); the head's K and V in LDS
//                  (rows padded to 36 floats); online softmax per lane over its eighth of the keys, merged by three lane
//                  exchanges; saves the row's
//                  log-sum-exp for the backward.
//   backward dq  : the same decomposition; p = exp(s - lse) recomputed, dS = p (dO.v - D) with D = dO.o.
//   backward dkv : workgroup = (32 keys, head, image), 8 lanes per key over eighths of the QUERY rows, the head's Q and
//                  dO (and lse, D) in LDS.
// Every output element is written exactly once (no zero fill, no atomics).  N <= 512 rows.
#include "common.h"

namespace sdetr {

constexpr int kAtD = 32, kAtRow = 36, kAtMaxN = 512, kAtThreads = 256;
constexpr int kAtLanes = 8;                                // lanes per row: each walks 1 / 8 of the keys (or of the query rows)
constexpr int kAtRowsPerBlock = kAtThreads / kAtLanes;    // 32
constexpr int kAtOwn = kAtD / kAtLanes;                    // channels a lane writes: 4

struct AtArgs {
    const float *q, *k, *v;       // element (b, n, h, c) at base + b * bs + n * rs + h * 32 + c
    int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs;
    float *o;                     // [B, N, H * 32]
    float *lse;                   // [B, H, N]
    const float *go;              // grad of o, [B, N, H * 32]
    float *gq, *gk, *gv;          // strides as q / k / v
    int B, H, N;
    float scale;
};

__device__ __forceinline__ void at_load_row(const float *src, float (&r)[kAtD])
{
#pragma unroll
    for (int i = 0; i < kAtD / 4; ++i) {
        const float4 v = reinterpret_cast<const float4 *>(src)[i];
        r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
    }
}
__device__ __forceinline__ float at_dot(const float (&a)[kAtD], const float *lds_row)
{
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kAtD / 4; ++i) {
        const float4 v = reinterpret_cast<const float4 *>(lds_row)[i];
        s = fmaf(a[4 * i], v.x, s); s = fmaf(a[4 * i + 1], v.y, s); s = fmaf(a[4 * i + 2], v.z, s); s = fmaf(a[4 * i + 3], v.w, s);
    }
    return s;
}
__device__ __forceinline__ void at_axpy(float (&acc)[kAtD], float a, const float *lds_row)
{
#pragma unroll
    for (int i = 0; i < kAtD / 4; ++i) {
        const float4 v = reinterpret_cast<const float4 *>(lds_row)[i];
        acc[4 * i] = fmaf(a, v.x, acc[4 * i]); acc[4 * i + 1] = fmaf(a, v.y, acc[4 * i + 1]);
        acc[4 * i + 2] = fmaf(a, v.z, acc[4 * i + 2]); acc[4 * i + 3] = fmaf(a, v.w, acc[4 * i + 3]);
    }
}
// rows [0, N) of a head's matrix (row stride rs floats in global memory) into LDS rows of kAtRow floats
// (four loads in flight per thread and round: one at a time the copy of a 300 x 32 matrix is ten dependent trips to memory)
__device__ __forceinline__ void at_stage(const float *src, int64_t rs, int N, float *dst, int tid)
{
    const int total = N * (kAtD / 4);
    for (int t0 = tid; t0 < total; t0 += 4 * kAtThreads) {
        float4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = min(t0 + u * kAtThreads, total - 1);
            r[u] = reinterpret_cast<const float4 *>(src + (int64_t)(t >> 3) * rs)[t & 7];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u * kAtThreads;
            if (t < total) reinterpret_cast<float4 *>(dst + (t >> 3) * kAtRow)[t & 7] = r[u];
        }
    }
}

__global__ void __launch_bounds__(kAtThreads) attention_train_fwd_kernel(AtArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float at_lds[];
    float *ks = at_lds, *vs = at_lds + p.N * kAtRow;
    const int tid = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
    at_stage(p.k + (int64_t)b * p.k_bs + h * kAtD, p.k_rs, p.N, ks, tid);
    at_stage(p.v + (int64_t)b * p.v_bs + h * kAtD, p.v_rs, p.N, vs, tid);
    __syncthreads();
    const int part = tid & (kAtLanes - 1), i = blockIdx.x * kAtRowsPerBlock + tid / kAtLanes;
    const bool live = i < p.N;
    const int ii = live ? i : p.N - 1;
    float q[kAtD], acc[kAtD];
    at_load_row(p.q + (int64_t)b * p.q_bs + (int64_t)ii * p.q_rs + h * kAtD, q);
#pragma unroll
    for (int c = 0; c < kAtD; ++c) { q[c] *= p.scale; acc[c] = 0.f; }
    const int per = (p.N + kAtLanes - 1) / kAtLanes, j0 = part * per, j1 = min(p.N, j0 + per);
    float m = -INFINITY, l = 0.f;
    for (int j = j0; j < j1; ++j) {
        const float s = at_dot(q, ks + j * kAtRow);
        if (s > m) {   // (rare after the first few keys)
            const float corr = expf(m - s);
            l *= corr;
#pragma unroll
            for (int c = 0; c < kAtD; ++c) acc[c] *= corr;
            m = s;
        }
        const float e = expf(s - m);
        l += e;
        at_axpy(acc, e, vs + j * kAtRow);
    }
    // the parts of a row sit in adjacent lanes
    float m_all = m;
#pragma unroll
    for (int o = 1; o < kAtLanes; o <<= 1) m_all = fmaxf(m_all, __shfl_xor(m_all, o, kAtLanes));
    const float w = m == -INFINITY ? 0.f : expf(m - m_all);   // (an empty part: fewer keys than lanes)
    l *= w;
#pragma unroll
    for (int o = 1; o < kAtLanes; o <<= 1) l += __shfl_xor(l, o, kAtLanes);
    const float inv = 1.0f / l;
    float mine[kAtOwn];
#pragma unroll
    for (int c = 0; c < kAtD; ++c) {
        float a = acc[c] * w;
#pragma unroll
        for (int o = 1; o < kAtLanes; o <<= 1) a += __shfl_xor(a, o, kAtLanes);
        if (c / kAtOwn == part) mine[c % kAtOwn] = a * inv;   // (c is a compile-time index: a select, not an indexed store)
    }
    if (live) {
        float *o = p.o + ((int64_t)b * p.N + i) * (p.H * kAtD) + h * kAtD + kAtOwn * part;
        reinterpret_cast<float4 *>(o)[0] = make_float4(mine[0], mine[1], mine[2], mine[3]);
        if (part == 0) p.lse[((int64_t)b * p.H + h) * p.N + i] = m_all + logf(l);
    }
}

__global__ void __launch_bounds__(kAtThreads) attention_train_bwd_dq_kernel(AtArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float at_lds[];
    float *ks = at_lds, *vs = at_lds + p.N * kAtRow;
    const int tid = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
    at_stage(p.k + (int64_t)b * p.k_bs + h * kAtD, p.k_rs, p.N, ks, tid);
    at_stage(p.v + (int64_t)b * p.v_bs + h * kAtD, p.v_rs, p.N, vs, tid);
    __syncthreads();
    const int part = tid & (kAtLanes - 1), i = blockIdx.x * kAtRowsPerBlock + tid / kAtLanes;
    const bool live = i < p.N;
    const int ii = live ? i : p.N - 1;
    float q[kAtD], go[kAtD], dq[kAtD];
    at_load_row(p.q + (int64_t)b * p.q_bs + (int64_t)ii * p.q_rs + h * kAtD, q);
    const int64_t orow = ((int64_t)b * p.N + ii) * (p.H * kAtD) + h * kAtD;
    at_load_row(p.go + orow, go);
    float D = 0.f;
    {
        float o[kAtD];
        at_load_row(p.o + orow, o);
#pragma unroll
        for (int c = 0; c < kAtD; ++c) D = fmaf(go[c], o[c], D);
    }
    const float lse = p.lse[((int64_t)b * p.H + h) * p.N + ii];
#pragma unroll
    for (int c = 0; c < kAtD; ++c) { q[c] *= p.scale; dq[c] = 0.f; }
    const int per = (p.N + kAtLanes - 1) / kAtLanes, j0 = part * per, j1 = min(p.N, j0 + per);
    for (int j = j0; j < j1; ++j) {
        const float pr = expf(at_dot(q, ks + j * kAtRow) - lse);
        const float ds = pr * (at_dot(go, vs + j * kAtRow) - D) * p.scale;
        at_axpy(dq, ds, ks + j * kAtRow);
    }
    float mine[kAtOwn];
#pragma unroll
    for (int c = 0; c < kAtD; ++c) {
        float a = dq[c];
#pragma unroll
        for (int o = 1; o < kAtLanes; o <<= 1) a += __shfl_xor(a, o, kAtLanes);
        if (c / kAtOwn == part) mine[c % kAtOwn] = a;
    }
    if (live) {
        float *g = p.gq + (int64_t)b * p.q_bs + (int64_t)i * p.q_rs + h * kAtD + kAtOwn * part;
        reinterpret_cast<float4 *>(g)[0] = make_float4(mine[0], mine[1], mine[2], mine[3]);
    }
}

__global__ void __launch_bounds__(kAtThreads) attention_train_bwd_dkv_kernel(AtArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float at_lds[];
    float *qs = at_lds, *gs = at_lds + p.N * kAtRow, *lse_s = gs + p.N * kAtRow, *d_s = lse_s + p.N;
    const int tid = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
    at_stage(p.q + (int64_t)b * p.q_bs + h * kAtD, p.q_rs, p.N, qs, tid);
    at_stage(p.go + (int64_t)b * p.N * (p.H * kAtD) + h * kAtD, p.H * kAtD, p.N, gs, tid);
    for (int n = tid; n < p.N; n += kAtThreads) {
        float go[kAtD], o[kAtD];
        const int64_t orow = ((int64_t)b * p.N + n) * (p.H * kAtD) + h * kAtD;
        at_load_row(p.go + orow, go);
        at_load_row(p.o + orow, o);
        float D = 0.f;
#pragma unroll
        for (int c = 0; c < kAtD; ++c) D = fmaf(go[c], o[c], D);
        d_s[n] = D;
        lse_s[n] = p.lse[((int64_t)b * p.H + h) * p.N + n];
    }
    __syncthreads();
    const int part = tid & (kAtLanes - 1), j = blockIdx.x * kAtRowsPerBlock + tid / kAtLanes;
    const bool live = j < p.N;
    const int jj = live ? j : p.N - 1;
    float k[kAtD], v[kAtD], dk[kAtD], dv[kAtD];
    at_load_row(p.k + (int64_t)b * p.k_bs + (int64_t)jj * p.k_rs + h * kAtD, k);
    at_load_row(p.v + (int64_t)b * p.v_bs + (int64_t)jj * p.v_rs + h * kAtD, v);
#pragma unroll
    for (int c = 0; c < kAtD; ++c) { k[c] *= p.scale; dk[c] = 0.f; dv[c] = 0.f; }
    const int per = (p.N + kAtLanes - 1) / kAtLanes, i0 = part * per, i1 = min(p.N, i0 + per);
    for (int i = i0; i < i1; ++i) {
        const float pr = expf(at_dot(k, qs + i * kAtRow) - lse_s[i]);
        const float ds = pr * (at_dot(v, gs + i * kAtRow) - d_s[i]) * p.scale;
        at_axpy(dk, ds, qs + i * kAtRow);
        at_axpy(dv, pr, gs + i * kAtRow);
    }
    float mk[kAtOwn], mv[kAtOwn];
#pragma unroll
    for (int c = 0; c < kAtD; ++c) {
        float a = dk[c], e = dv[c];
#pragma unroll
        for (int o = 1; o < kAtLanes; o <<= 1) { a += __shfl_xor(a, o, kAtLanes); e += __shfl_xor(e, o, kAtLanes); }
        if (c / kAtOwn == part) { mk[c % kAtOwn] = a; mv[c % kAtOwn] = e; }
    }
    if (live) {
        float *g = p.gk + (int64_t)b * p.k_bs + (int64_t)j * p.k_rs + h * kAtD + kAtOwn * part;
        reinterpret_cast<float4 *>(g)[0] = make_float4(mk[0], mk[1], mk[2], mk[3]);
        float *gv = p.gv + (int64_t)b * p.v_bs + (int64_t)j * p.v_rs + h * kAtD + kAtOwn * part;
        reinterpret_cast<float4 *>(gv)[0] = make_float4(mv[0], mv[1], mv[2], mv[3]);
    }
}

}  // namespace sdetr

using namespace sdetr;

static int at_check(int B, int H, int N, int head_dim, const AtArgs &a)
{
    if (head_dim != kAtD) return fail("attention_train: built for 32-channel heads (got %d)", head_dim);
    if (B < 0 || H <= 0 || N < 0) return fail("attention_train: bad sizes");
    if (N > kAtMaxN) return fail("attention_train: at most %d rows (got %d)", kAtMaxN, N);
    const int64_t w = (int64_t)H * kAtD;
    if (a.q_rs < w || a.k_rs < w || a.v_rs < w || (a.q_rs & 3) || (a.k_rs & 3) || (a.v_rs & 3) || (a.q_bs & 3) || (a.k_bs & 3) ||
        (a.v_bs & 3))
        return fail("attention_train: row strides must cover the heads and be multiples of 4 floats");
    return 0;
}

extern "C" int sdetr_attention_train_max_rows(void) { return kAtMaxN; }

extern "C" int sdetr_attention_train_forward_f32(sdetr_stream_t stream, const float *q, int64_t q_batch_stride, int64_t q_row_stride,
                                                 const float *k, int64_t k_batch_stride, int64_t k_row_stride, const float *v,
                                                 int64_t v_batch_stride, int64_t v_row_stride, int batch_size, int num_heads,
                                                 int num_rows, int head_dim, float scale, float *out, float *lse)
{
    AtArgs a{};
    a.q = q; a.k = k; a.v = v; a.q_bs = q_batch_stride; a.q_rs = q_row_stride; a.k_bs = k_batch_stride; a.k_rs = k_row_stride;
    a.v_bs = v_batch_stride; a.v_rs = v_row_stride; a.o = out; a.lse = lse; a.B = batch_size; a.H = num_heads; a.N = num_rows;
    a.scale = scale;
    if (int e = at_check(batch_size, num_heads, num_rows, head_dim, a)) return e;
    if ((int64_t)batch_size * num_rows == 0) return 0;
    if (!q || !k || !v || !out || !lse) return fail("attention_train: null pointer");
    const size_t lds = (size_t)2 * num_rows * kAtRow * 4;
    static DeviceOnce once;
    allow_dynamic_lds(attention_train_fwd_kernel, once, 2 * kAtMaxN * kAtRow * 4);
    const dim3 grid((unsigned)((num_rows + kAtRowsPerBlock - 1) / kAtRowsPerBlock), (unsigned)num_heads, (unsigned)batch_size);
    hipLaunchKernelGGL(attention_train_fwd_kernel, grid, dim3(kAtThreads), lds, static_cast<hipStream_t>(stream), a);
    return check_launch("attention_train_forward");
}

extern "C" int sdetr_attention_train_backward_f32(sdetr_stream_t stream, const float *q, int64_t q_batch_stride, int64_t q_row_stride,
                                                  const float *k, int64_t k_batch_stride, int64_t k_row_stride, const float *v,
                                                  int64_t v_batch_stride, int64_t v_row_stride, int batch_size, int num_heads,
                                                  int num_rows, int head_dim, float scale, const float *out, const float *lse,
                                                  const float *grad_out, float *grad_q, float *grad_k, float *grad_v)
{
    AtArgs a{};
    a.q = q; a.k = k; a.v = v; a.q_bs = q_batch_stride; a.q_rs = q_row_stride; a.k_bs = k_batch_stride; a.k_rs = k_row_stride;
    a.v_bs = v_batch_stride; a.v_rs = v_row_stride; a.o = const_cast<float *>(out); a.lse = const_cast<float *>(lse);
    a.go = grad_out; a.gq = grad_q; a.gk = grad_k; a.gv = grad_v; a.B = batch_size; a.H = num_heads; a.N = num_rows; a.scale = scale;
    if (int e = at_check(batch_size, num_heads, num_rows, head_dim, a)) return e;
    if ((int64_t)batch_size * num_rows == 0) return 0;
    if (!q || !k || !v || !out || !lse || !grad_out || !grad_q || !grad_k || !grad_v) return fail("attention_train: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((num_rows + kAtRowsPerBlock - 1) / kAtRowsPerBlock), (unsigned)num_heads, (unsigned)batch_size);
    static DeviceOnce once_q, once_kv;
    allow_dynamic_lds(attention_train_bwd_dq_kernel, once_q, 2 * kAtMaxN * kAtRow * 4);
    allow_dynamic_lds(attention_train_bwd_dkv_kernel, once_kv, (2 * kAtMaxN * kAtRow + 2 * kAtMaxN) * 4);
    hipLaunchKernelGGL(attention_train_bwd_dq_kernel, grid, dim3(kAtThreads), (size_t)2 * num_rows * kAtRow * 4, s, a);
    if (int e = check_launch("attention_train_backward_dq")) return e;
    hipLaunchKernelGGL(attention_train_bwd_dkv_kernel, grid, dim3(kAtThreads), ((size_t)2 * num_rows * kAtRow + 2 * num_rows) * 4, s, a);
    return check_launch("attention_train_backward_dkv");
}
