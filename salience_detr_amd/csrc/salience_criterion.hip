// Salience supervision (row N4; reference models/detectors/salience_detr.py:13-116, models/bricks/losses.py:4-13).
//
//  * salience_target_kernel: get_mask_single_level for every level and image in one launch.  One thread per (image,
//    token): pixel centre ((x + 0.5) * stride_x, (y + 0.5) * stride_y), and over the image's ground-truth boxes (staged
//    in LDS) the scale-independent confidence 1 - sqrt(dx^2 + dy^2) / 2 of the boxes that contain the centre (max over
//    boxes), kept only if one containing box has its largest border distance inside the level's range
//    (limit_range).  The reference builds [h*w, m, 4] distance tensors per (level, image) with ~25 launches each.
//  * focal_loss_kernel: sigmoid focal loss with a weight that keeps its gradient; per-element loss summed and the
//    positives counted by a deterministic two-stage reduction (block partials, then one block), the loss
//    sum / max(#positives, 1) written by the second stage.  (loss.sum(1) / S summed, / num_pos, * S: S cancels.)
//  * focal_grad_kernel: d loss / d logit = upstream * (w (p - t) + bce * dw/dx) / num_pos.
#include "common.h"

namespace sdetr {

constexpr int kCritMaxLevels = 8;
constexpr int kCritBoxChunk = 256;   // boxes staged per pass

struct TargetArgs {
    const float *boxes;        // [sum m, 4] xyxy in input-image pixels
    const int *box_offset;     // [B + 1]
    int height[kCritMaxLevels], width[kCritMaxLevels], start[kCritMaxLevels];
    float stride_y[kCritMaxLevels], stride_x[kCritMaxLevels], lo[kCritMaxLevels], hi[kCritMaxLevels];
    int L, S;
    float noise_scale;
    const float *noise;        // [B, S] or NULL
    float *target;             // [B, S]
};

__global__ void __launch_bounds__(256) salience_target_kernel(TargetArgs p)
{
    __shared__ float4 sb[kCritBoxChunk];
    const int b = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    const bool live = s < p.S;
    int lvl = 0;
    while (lvl + 1 < p.L && s >= p.start[lvl + 1]) ++lvl;
    const int sp = s - p.start[lvl];
    const int y = sp / p.width[lvl], x = sp - y * p.width[lvl];
    // torch.linspace(0.5, n - 0.5, n) * stride: the linspace values are exactly i + 0.5
    const float cx = ((float)x + 0.5f) * p.stride_x[lvl], cy = ((float)y + 0.5f) * p.stride_y[lvl];
    const float lo = p.lo[lvl], hi = p.hi[lvl];
    const int b0 = p.box_offset[b], b1 = p.box_offset[b + 1];
    float best = 0.f;
    bool pos = false, any = false;
    for (int base = b0; base < b1; base += kCritBoxChunk) {
        const int n = min(kCritBoxChunk, b1 - base);
        __syncthreads();
        if ((int)threadIdx.x < n) sb[threadIdx.x] = reinterpret_cast<const float4 *>(p.boxes)[base + threadIdx.x];
        __syncthreads();
        if (!live) continue;
        for (int i = 0; i < n; ++i) {
            const float4 g = sb[i];
            const float l = cx - g.x, t = cy - g.y, r = g.z - cx, bt = g.w - cy;
            const float dmin = fminf(fminf(l, t), fminf(r, bt)), dmax = fmaxf(fmaxf(l, t), fmaxf(r, bt));
            if (!(dmin > 0.f)) continue;                       // (confidence_per_box[~mask_in_gt_boxes] = 0)
            const float dx = (l - r) / (l + r), dy = (t - bt) / (t + bt);
            const float conf = 1.f - sqrtf(dx * dx + dy * dy) / 2.f;
            best = any ? fmaxf(best, conf) : fmaxf(conf, 0.f); // max over all boxes, the others contribute 0
            any = true;
            pos |= dmax > lo && dmax <= hi;
        }
    }
    if (!live) return;
    float v = pos ? best : 0.f;
    if (p.noise_scale != 0.f) v = (1.f - p.noise_scale) * v + p.noise_scale * p.noise[(int64_t)b * p.S + s];
    p.target[(int64_t)b * p.S + s] = v;
}

__device__ __forceinline__ void focal_terms(float x, float t, float alpha, float gamma, float &loss, float &dloss)
{
    const float p = 1.f / (1.f + expf(-x));
    const float q = 1.f - p;
    // binary_cross_entropy_with_logits: max(x, 0) - x t + log(1 + exp(-|x|))
    const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    float pg, qg, dpg, dqg;   // p^gamma, q^gamma and their derivatives with respect to x
    if (gamma == 2.f) {
        pg = p * p; qg = q * q;
        dpg = 2.f * p * p * q; dqg = -2.f * q * q * p;
    } else {
        pg = powf(p, gamma); qg = powf(q, gamma);
        dpg = gamma * pg * q; dqg = -gamma * qg * p;
    }
    const float w = (1.f - alpha) * pg * (1.f - t) + t * alpha * qg;
    const float dw = (1.f - alpha) * dpg * (1.f - t) + t * alpha * dqg;
    loss = bce * w;
    dloss = w * (p - t) + bce * dw;
}

struct LossArgs {
    const float *logits;   // [n]
    const float *target;   // [n]
    int64_t n;
    float alpha, gamma, pos_threshold;
    float *partial;        // [2 * blocks]: loss sums, positive counts
    float *out;            // [2]: loss, num_pos
};

__device__ __forceinline__ float block_sum_256(float v, float *scratch)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

__global__ void __launch_bounds__(256) focal_loss_kernel(LossArgs p)
{
    __shared__ float scratch[4];
    float ls = 0.f, np = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * 256) {
        float l, d;
        const float t = p.target[i];
        focal_terms(p.logits[i], t, p.alpha, p.gamma, l, d);
        ls += l;
        np += t > p.pos_threshold ? 1.f : 0.f;
    }
    const float a = block_sum_256(ls, scratch);
    const float c = block_sum_256(np, scratch);
    if (threadIdx.x == 0) { p.partial[blockIdx.x] = a; p.partial[gridDim.x + blockIdx.x] = c; }
}

__global__ void __launch_bounds__(256) focal_finish_kernel(const float *partial, int blocks, float *out)
{
    __shared__ float scratch[4];
    float ls = 0.f, np = 0.f;
    for (int i = threadIdx.x; i < blocks; i += 256) { ls += partial[i]; np += partial[blocks + i]; }
    const float a = block_sum_256(ls, scratch);
    const float c = block_sum_256(np, scratch);
    if (threadIdx.x == 0) {
        const float num_pos = fmaxf(c, 1.f);
        out[0] = a / num_pos;
        out[1] = num_pos;
    }
}

__global__ void __launch_bounds__(256) focal_grad_kernel(const float *logits, const float *target, int64_t n, float alpha,
                                                         float gamma, const float *loss_and_num_pos, const float *upstream,
                                                         float *grad)
{
    const float scale = upstream[0] / loss_and_num_pos[1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float l, d;
        focal_terms(logits[i], target[i], alpha, gamma, l, d);
        grad[i] = d * scale;
    }
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int sdetr_salience_targets(sdetr_stream_t stream, const float *boxes_xyxy, const int *box_offset, int batch_size,
                                      const int64_t *level_shapes_host, const float *level_strides_host,
                                      const float *limit_range_host, int num_levels, float noise_scale,
                                      const float *noise, float *target)
{
    if (num_levels <= 0 || num_levels > kCritMaxLevels || batch_size < 0) return fail("salience_targets: bad sizes");
    if (!level_shapes_host || !level_strides_host || !limit_range_host) return fail("salience_targets: null level description");
    TargetArgs a{};
    int64_t cur = 0;
    for (int l = 0; l < num_levels; ++l) {
        a.height[l] = (int)level_shapes_host[2 * l]; a.width[l] = (int)level_shapes_host[2 * l + 1];
        if (a.height[l] <= 0 || a.width[l] <= 0) return fail("salience_targets: empty level");
        a.stride_y[l] = level_strides_host[2 * l]; a.stride_x[l] = level_strides_host[2 * l + 1];
        a.lo[l] = limit_range_host[2 * l]; a.hi[l] = limit_range_host[2 * l + 1];
        a.start[l] = (int)cur;
        cur += (int64_t)a.height[l] * a.width[l];
    }
    if (cur > 0x7fffffff) return fail("salience_targets: pyramid too large");
    if (batch_size == 0 || cur == 0) return 0;
    if (!box_offset || !target || (noise_scale != 0.f && !noise)) return fail("salience_targets: null pointer");
    a.boxes = boxes_xyxy; a.box_offset = box_offset; a.L = num_levels; a.S = (int)cur; a.noise_scale = noise_scale;
    a.noise = noise; a.target = target;
    hipLaunchKernelGGL(salience_target_kernel, dim3((unsigned)((cur + 255) / 256), batch_size), dim3(256), 0,
                       (hipStream_t)stream, a);
    return check_launch("salience_targets");
}

extern "C" int64_t sdetr_focal_loss_workspace_bytes(int64_t count)
{
    const int64_t blocks = count > 0 ? (count + 4095) / 4096 : 0;
    return (blocks > 1024 ? 1024 : blocks) * 2 * (int64_t)sizeof(float);
}

extern "C" int sdetr_salience_focal_loss(sdetr_stream_t stream, const float *logits, const float *target, int64_t count,
                                         float alpha, float gamma, float positive_threshold, void *workspace,
                                         int64_t workspace_bytes, float *loss_and_num_pos)
{
    if (count < 0) return fail("salience_focal_loss: negative count");
    if (!loss_and_num_pos) return fail("salience_focal_loss: null output");
    int64_t blocks = (count + 4095) / 4096;
    if (blocks > 1024) blocks = 1024;
    if (blocks == 0) blocks = 1;
    if (!workspace || workspace_bytes < blocks * 2 * (int64_t)sizeof(float)) return fail("salience_focal_loss: workspace too small");
    if (count > 0 && (!logits || !target)) return fail("salience_focal_loss: null pointer");
    LossArgs a{logits, target, count, alpha, gamma, positive_threshold, (float *)workspace, loss_and_num_pos};
    hipLaunchKernelGGL(focal_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(focal_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float *)workspace, (int)blocks,
                       loss_and_num_pos);
    return check_launch("salience_focal_loss");
}

extern "C" int sdetr_salience_focal_loss_backward(sdetr_stream_t stream, const float *logits, const float *target,
                                                  int64_t count, float alpha, float gamma, const float *loss_and_num_pos,
                                                  const float *grad_loss, float *grad_logits)
{
    if (count < 0) return fail("salience_focal_loss_backward: negative count");
    if (count == 0) return 0;
    if (!logits || !target || !loss_and_num_pos || !grad_loss || !grad_logits) return fail("salience_focal_loss_backward: null pointer");
    int64_t blocks = (count + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(focal_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, target, count, alpha,
                       gamma, loss_and_num_pos, grad_loss, grad_logits);
    return check_launch("salience_focal_loss_backward");
}
