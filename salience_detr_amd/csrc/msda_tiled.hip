// Multi-scale deformable attention forward with the value maps staged through LDS (gfx950).
//
// Why: the direct gather (msda_forward.hip) is bound by the L2 -> L1 line-fill path.  At the
// benchmark shape every (pixel, head) of the value pyramid is read ~32 times per launch (11 363
// queries x 16 samples x 4 corners over 22 323 pixels), yet the 32 KiB L1 keeps almost none of it
// (measured: identical time for spatially sorted and random query order, ~15 TB/s of line fills
// for 0.08 GB of algorithmic bytes).  Here the reuse is made explicit:
//
//  * queries are bucketed by the TY x TX level-0 region their reference point falls in
//    (region_bucket_kernel: LDS histogram + scan + scatter, one workgroup per image);
//  * one workgroup = (image, head, region).  It derives, per level, the bounding box of its
//    queries' reference points, pads it by HALO pixels, and loads that window of the head-major
//    bf16 value map into LDS ONCE with coalesced 16-byte loads (out-of-image pixels are stored as
//    zeros, which IS the reference's zero padding -- no per-corner validity tests remain);
//  * the bilinear gather then runs out of LDS: a quad of 4 lanes owns one (query, head) row, each
//    lane 8 channels = one ds_read_b128 per corner; descriptors (two packed LDS row offsets + four
//    weights) are computed once per sample by one lane of the quad (lane j owns level j: L = P = 4)
//    and broadcast through a wave-private LDS slab -- no workgroup barrier in the sample loop;
//  * a sample that falls outside its level's window (offset larger than HALO) takes a per-sample
//    global-memory fallback inside the same kernel, so the result never depends on the window.
//
// blockIdx % num_heads is the head, so (block b -> XCD b % 8) every XCD's L2 serves one head's slab.
// Supported shape: D = 32 bf16 head-major value, L = 4, P = 4 (every Salience-DETR config); anything
// else is routed to the direct kernel by the host wrapper.
#include "common.h"
#include "tiled_geometry.h"

namespace sdetr {

void note_forward_kernel(int which);  // abi.hip

constexpr int kPixBytes = kTD * 2;           // bf16
constexpr int kTileBytes = kTilePx * kPixBytes;          // 65 280
constexpr int kDescW = 4 * 8 * 16 * 16;                  // [wave][sample][row] float4
constexpr int kDescO = 4 * 8 * 16 * 4;                   // [wave][sample][row] u32
constexpr int kTiledLds = kTileBytes + kDescW + kDescO;
static_assert(kTiledLds <= 80 * 1024, "two workgroups per CU");
static_assert(kTiledLds / 16 < 65536, "16-bit LDS offsets");

// ------------------------------------------------------------------------------------------------
// Region bucketing: order[b] = query slots grouped by region, region_start[b] = CSR offsets,
// region_box[b][r][l] = {min x, min y, max x, max y} of floor(ref_l * size_l - 0.5) over the region's
// queries (the zero-offset sample pixel on level l) -- everything the gather workgroup needs to place
// its windows arrives with ONE scalar load instead of a dependent ord -> ref -> reduce chain.
struct BucketArgs {
    // position source: pos[q*q_stride + h*head_stride + l*lvl_stride + {0,1}], averaged over n_heads heads.
    // reference points [B,Nq,L,ref_dim]: q_stride = L*ref_dim, lvl_stride = ref_dim, n_heads = 1;
    // sampling locations [B,Nq,M,L,P,2] (backward op, no reference points at hand): q_stride = M*L*P*2,
    // head_stride = L*P*2, lvl_stride = P*2, n_heads = M -- the mean of point 0 over the heads is the
    // reference point up to the (direction-symmetric) learned offsets.
    const float *ref;
    int64_t q_stride;
    int head_stride, lvl_stride, n_heads;
    const int64_t *shapes;
    int B, Nq, L, H0, W0, RX, RY;
    int32_t *order;         // [B,Nq]
    int32_t *region_start;  // [B,RX*RY+1]
    int32_t *region_box;    // [B,RX*RY,kTL,4]
};

__device__ __forceinline__ float2 bucket_pos(const BucketArgs &p, int b, int q, int l)
{
    const float *r = p.ref + ((int64_t)b * p.Nq + q) * p.q_stride + l * p.lvl_stride;
    float x = 0.f, y = 0.f;
    for (int h = 0; h < p.n_heads; ++h) {
        x += r[h * p.head_stride];
        y += r[h * p.head_stride + 1];
    }
    const float inv = 1.f / (float)p.n_heads;
    return make_float2(x * inv, y * inv);
}

__device__ __forceinline__ int region_of(const BucketArgs &p, int b, int q)
{
    const float2 r = bucket_pos(p, b, q, 0);
    int rx = (int)floorf(r.x * (float)p.W0) / kTX;
    int ry = (int)floorf(r.y * (float)p.H0) / kTY;
    rx = rx < 0 ? 0 : (rx >= p.RX ? p.RX - 1 : rx);
    ry = ry < 0 ? 0 : (ry >= p.RY ? p.RY - 1 : ry);
    return ry * p.RX + rx;
}

__global__ void __launch_bounds__(1024) region_bucket_kernel(BucketArgs p)
{
    extern __shared__ int bsm[];  // [R] counts -> cursors | [16] scan partials | [R][kTL][4] boxes
    const int R = p.RX * p.RY;
    int *cnt = bsm;
    int *part = bsm + R;
    int *box = bsm + R + 16;
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int r = tid; r < R; r += 1024) cnt[r] = 0;
    for (int i = tid; i < R * kTL * 4; i += 1024) box[i] = (i & 2) ? -(1 << 30) : (1 << 30);
    __syncthreads();
    for (int q = tid; q < p.Nq; q += 1024) atomicAdd(&cnt[region_of(p, b, q)], 1);
    __syncthreads();
    // exclusive scan of cnt[0..R): each thread owns a contiguous run
    const int per = (R + 1023) / 1024;
    const int lo = tid * per, hi = min(R, lo + per);
    int local = 0;
    for (int r = lo; r < hi; ++r) local += cnt[r];
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(incl, o, 64);
        if ((tid & 63) >= o) incl += n;
    }
    if ((tid & 63) == 63) part[tid >> 6] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < (tid >> 6); ++w) before += part[w];
    int run = before + incl - local;
    int32_t *rs = p.region_start + (int64_t)b * (R + 1);
    for (int r = lo; r < hi; ++r) {
        const int c = cnt[r];
        rs[r] = run;
        cnt[r] = run;  // becomes the scatter cursor
        run += c;
    }
    if (tid == 0) rs[R] = p.Nq;
    __syncthreads();
    for (int q = tid; q < p.Nq; q += 1024) {
        const int rid = region_of(p, b, q);
        const int pos = atomicAdd(&cnt[rid], 1);
        p.order[(int64_t)b * p.Nq + pos] = q;
#pragma unroll
        for (int l = 0; l < kTL; ++l) {
            const float2 r = bucket_pos(p, b, q, l);
            const int ix = (int)floorf(r.x * (float)p.shapes[2 * l + 1] - 0.5f);
            const int iy = (int)floorf(r.y * (float)p.shapes[2 * l] - 0.5f);
            int *bx = box + (rid * kTL + l) * 4;
            atomicMin(&bx[0], ix);
            atomicMin(&bx[1], iy);
            atomicMax(&bx[2], ix);
            atomicMax(&bx[3], iy);
        }
    }
    __syncthreads();
    int32_t *gb = p.region_box + (int64_t)b * R * kTL * 4;
    for (int i = tid; i < R * kTL * 4; i += 1024) gb[i] = box[i];
}

// ------------------------------------------------------------------------------------------------
struct TiledArgs {
    const char *value;  // [B,M,Nv,32] bf16
    const int64_t *shapes;
    const int64_t *lsi;
    const float *ref;
    int ref_dim;
    const void *proj;
    int proj_bf16;
    int64_t proj_stride;
    const int32_t *order;
    const int32_t *region_start;
    const int32_t *region_box;
    int R;
    void *out;
    int out_bf16;
    int B, Nv, M, Nq;
};

__device__ __forceinline__ void fma8(float *acc, const uint4 &v, float w)
{
    acc[0] = fmaf(w, bf16_lo(v.x), acc[0]);
    acc[1] = fmaf(w, bf16_hi(v.x), acc[1]);
    acc[2] = fmaf(w, bf16_lo(v.y), acc[2]);
    acc[3] = fmaf(w, bf16_hi(v.y), acc[3]);
    acc[4] = fmaf(w, bf16_lo(v.z), acc[4]);
    acc[5] = fmaf(w, bf16_hi(v.z), acc[5]);
    acc[6] = fmaf(w, bf16_lo(v.w), acc[6]);
    acc[7] = fmaf(w, bf16_hi(v.w), acc[7]);
}

// one row's raw inputs for one lane (level j of a quad): 4 offsets (x,y), 4 logits, the reference point
struct RowIn {
    uint4 o0, o1;  // offsets: bf16 -> o0 only (8 x bf16); f32 -> o0,o1 (8 floats)
    uint4 g;       // logits: bf16 -> g.x,g.y (4 x bf16); f32 -> 4 floats
    float4 r;      // reference point (x, y[, w, h])
};

__device__ __forceinline__ RowIn load_row(const TiledArgs &p, int64_t bq, int m, int j, bool active)
{
    // unconditional loads (the caller clamps an inactive row to query 0); `active` only masks later use
    (void)active;
    RowIn in;
    in.o1 = make_uint4(0u, 0u, 0u, 0u);
    in.g = make_uint4(0u, 0u, 0u, 0u);
    in.r = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int LP = kTL * kTP;
    const int64_t o_idx = bq * p.proj_stride + ((int64_t)m * LP + j * kTP) * 2;
    const int64_t l_idx = bq * p.proj_stride + (int64_t)p.M * LP * 2 + (int64_t)m * LP + j * kTP;
    if (p.proj_bf16) {
        const bf16_t *pp = reinterpret_cast<const bf16_t *>(p.proj);
        in.o0 = *reinterpret_cast<const uint4 *>(pp + o_idx);
        const uint2 g = *reinterpret_cast<const uint2 *>(pp + l_idx);
        in.g.x = g.x;
        in.g.y = g.y;
    } else {
        const float *pp = reinterpret_cast<const float *>(p.proj);
        in.o0 = *reinterpret_cast<const uint4 *>(pp + o_idx);
        in.o1 = *reinterpret_cast<const uint4 *>(pp + o_idx + 4);
        in.g = *reinterpret_cast<const uint4 *>(pp + l_idx);
    }
    const float *r = p.ref + (bq * kTL + j) * p.ref_dim;
    if (p.ref_dim == 4) {
        in.r = *reinterpret_cast<const float4 *>(r);
    } else {
        const float2 xy = *reinterpret_cast<const float2 *>(r);
        in.r.x = xy.x;
        in.r.y = xy.y;
    }
    return in;
}

constexpr int stage_passes(int l) { return (tile_w_cap(l) * 4 + 63) / 64; }
constexpr int stage_rows(int l) { return (tile_h_cap(l) + 3) / 4; }
static_assert(stage_passes(0) == 2 && stage_rows(0) == 5 && stage_passes(1) == 2 && stage_rows(1) == 4 &&
                  stage_passes(2) == 1 && stage_rows(2) == 3 && stage_passes(3) == 1 && stage_rows(3) == 3,
              "the SDETR_STAGE_* expansion list below is written for exactly this geometry");

// One window load.  Lane = 16-byte chunk within a window row (pass PS covers chunks 64*PS..), wavefront w takes
// rows w + 4*R: the row part of an address is wave-uniform, the lane part is computed once per (level, pass).
// Loads are unconditional (clamped to a safe address).  d encodes what to do with the value afterwards:
// >= 0 store it at that LDS offset, -1 skip, <= -2 store zeros at (-d - 2) (pixel outside the image).
// Named scalars instead of arrays: hipcc left the equivalent uint4[24] in scratch with a wait after every load.
#define SDETR_STAGE_LOAD(L, PS, R)                                                                              \
    uint4 sv_##L##_##PS##_##R;                                                                                  \
    int sd_##L##_##PS##_##R;                                                                                    \
    {                                                                                                           \
        const int c = lane + 64 * PS;                                                                           \
        const int x = c >> 2, part = c & 3;                                                                     \
        const int gx = OX[L] + x;                                                                               \
        const int y = wave + 4 * R;                                                                             \
        const int gy = OY[L] + y;                                                                               \
        const bool in_win = c < TW[L] * 4 && y < TH[L];                                                         \
        const bool in_img = in_win && gx >= 0 && gx < LW[L] && gy >= 0 && gy < LH[L];                           \
        const char *src = in_img ? vbase + (int64_t)(LS[L] + gy * LW[L] + gx) * kPixBytes + part * 16 : vbase;  \
        sv_##L##_##PS##_##R = *reinterpret_cast<const uint4 *>(src);                                            \
        const int dst = (tile_base_px(L) + y * tile_w_cap(L) + x) * kPixBytes + part * 16;                      \
        sd_##L##_##PS##_##R = in_win ? (in_img ? dst : -(dst + 2)) : -1;                                        \
    }
#define SDETR_STAGE_STORE(L, PS, R)                                                                             \
    {                                                                                                           \
        const int d = sd_##L##_##PS##_##R;                                                                      \
        if (d >= 0) *reinterpret_cast<uint4 *>(tile + d) = sv_##L##_##PS##_##R;                                 \
        else if (d != -1) *reinterpret_cast<uint4 *>(tile + (-d - 2)) = make_uint4(0u, 0u, 0u, 0u);             \
    }
#define SDETR_STAGE_ALL(OP)                                                                                     \
    OP(0, 0, 0) OP(0, 0, 1) OP(0, 0, 2) OP(0, 0, 3) OP(0, 0, 4) OP(0, 1, 0) OP(0, 1, 1) OP(0, 1, 2) OP(0, 1, 3)   \
    OP(0, 1, 4) OP(1, 0, 0) OP(1, 0, 1) OP(1, 0, 2) OP(1, 0, 3) OP(1, 1, 0) OP(1, 1, 1) OP(1, 1, 2) OP(1, 1, 3)   \
    OP(2, 0, 0) OP(2, 0, 1) OP(2, 0, 2) OP(3, 0, 0) OP(3, 0, 1) OP(3, 0, 2)

__global__ void __launch_bounds__(kBlock, 2) msda_tiled_kernel(TiledArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *tile = lds;
    float4 *dW = reinterpret_cast<float4 *>(lds + kTileBytes);
    uint32_t *dO = reinterpret_cast<uint32_t *>(lds + kTileBytes + kDescW);

    const int tid = threadIdx.x;
    const int m = blockIdx.x % p.M;
    const int br = blockIdx.x / p.M;
    const int b = br / p.R;
    const int region = br - b * p.R;
    // ---- workgroup-uniform facts: all scalar loads, no LDS table, no barrier ----
    const int rs = p.region_start[(int64_t)b * (p.R + 1) + region];
    const int nq = p.region_start[(int64_t)b * (p.R + 1) + region + 1] - rs;
    if (nq <= 0) return;
    const int32_t *ord = p.order + (int64_t)b * p.Nq + rs;
    int LW[kTL], LH[kTL], LS[kTL], OX[kTL], OY[kTL], TW[kTL], TH[kTL];
    {
        const int4 *boxes = reinterpret_cast<const int4 *>(p.region_box) + ((int64_t)b * p.R + region) * kTL;
#pragma unroll
        for (int l = 0; l < kTL; ++l) {
            LH[l] = (int)p.shapes[2 * l];
            LW[l] = (int)p.shapes[2 * l + 1];
            LS[l] = (int)p.lsi[l];
            const int4 bx = boxes[l];  // min x, min y, max x, max y
            OX[l] = bx.x - kHalo;
            OY[l] = bx.y - kHalo;
            TW[l] = min(tile_w_cap(l), bx.z - bx.x + 2 * kHalo + 2);
            TH[l] = min(tile_h_cap(l), bx.w - bx.y + 2 * kHalo + 2);
        }
    }

    const int lane = tid & 63, wave = tid >> 6;
    const int qd = lane >> 2, j = lane & 3;
    const int nchunks = (nq + 63) >> 6;

    // ---- chunk 0's query slot first (its projection row is a dependent load: start the chain now) ----
    int i_cur = wave * 16 + qd;
    int q_cur = i_cur < nq ? ord[i_cur] : -1;
    int q_nxt = (i_cur + 64) < nq ? ord[i_cur + 64] : -1;

    // ---- stage the four windows (zeros outside the image): ALL loads are issued before the first LDS
    // store, so the whole window is in flight at once (stage_issue) ----
    const char *vbase = p.value + ((int64_t)b * p.M + m) * p.Nv * (int64_t)kPixBytes;
    SDETR_STAGE_ALL(SDETR_STAGE_LOAD)
    // chunk 0's inputs: issued while the window loads are still in flight
    RowIn in_cur = load_row(p, (int64_t)b * p.Nq + max(q_cur, 0), m, j, q_cur >= 0);
    SDETR_STAGE_ALL(SDETR_STAGE_STORE)
    __syncthreads();

    // ---- rows: a quad per (query, head); lane j of the quad owns level j's four samples ----
    float4 *myW = dW + wave * (8 * 16);
    uint32_t *myO = dO + wave * (8 * 16);
    const int Wl = j == 0 ? LW[0] : j == 1 ? LW[1] : j == 2 ? LW[2] : LW[3];
    const int Hl = j == 0 ? LH[0] : j == 1 ? LH[1] : j == 2 ? LH[2] : LH[3];
    const int oxl = j == 0 ? OX[0] : j == 1 ? OX[1] : j == 2 ? OX[2] : OX[3];
    const int oyl = j == 0 ? OY[0] : j == 1 ? OY[1] : j == 2 ? OY[2] : OY[3];
    const int twl = j == 0 ? TW[0] : j == 1 ? TW[1] : j == 2 ? TW[2] : TW[3];
    const int thl = j == 0 ? TH[0] : j == 1 ? TH[1] : j == 2 ? TH[2] : TH[3];
    const uint32_t lvl_base16 = (uint32_t)((j == 0 ? tile_base_px(0) : j == 1 ? tile_base_px(1)
                                            : j == 2 ? tile_base_px(2) : tile_base_px(3)) * (kPixBytes / 16));
    const uint32_t pitch16 = (uint32_t)((j == 0 ? tile_w_cap(0) : j == 1 ? tile_w_cap(1)
                                         : j == 2 ? tile_w_cap(2) : tile_w_cap(3)) * (kPixBytes / 16));
    for (int c = 0; c < nchunks; ++c) {
        const bool active = q_cur >= 0;
        const int64_t bq = (int64_t)b * p.Nq + max(q_cur, 0);
        // software pipeline: next chunk's inputs (its slot arrived a chunk ago) and the slot after that
        const RowIn in_nxt = load_row(p, (int64_t)b * p.Nq + max(q_nxt, 0), m, j, q_nxt >= 0);
        const int i_nn = i_cur + 128;
        const int q_nn = (c + 2 < nchunks && i_nn < nq) ? ord[i_nn] : -1;

        // sample t of this lane: raw offset / logit straight out of the input registers (recomputed by the
        // rare fallback path instead of being kept live across the LDS rounds)
        auto raw = [&](int t, float &oxv, float &oyv, float &lgv) {
            if (p.proj_bf16) {
                const uint32_t o = t == 0 ? in_cur.o0.x : t == 1 ? in_cur.o0.y : t == 2 ? in_cur.o0.z : in_cur.o0.w;
                const uint32_t g = t < 2 ? in_cur.g.x : in_cur.g.y;
                oxv = bf16_lo(o);
                oyv = bf16_hi(o);
                lgv = (t & 1) ? bf16_hi(g) : bf16_lo(g);
            } else {
                const uint4 &o = t < 2 ? in_cur.o0 : in_cur.o1;
                oxv = __uint_as_float((t & 1) ? o.z : o.x);
                oyv = __uint_as_float((t & 1) ? o.w : o.y);
                lgv = __uint_as_float(t == 0 ? in_cur.g.x : t == 1 ? in_cur.g.y : t == 2 ? in_cur.g.z : in_cur.g.w);
            }
        };
        // softmax over the quad's 16 logits
        float mx, inv;
        {
            float l0, l1, l2, l3, d0, d1;
            raw(0, d0, d1, l0); raw(1, d0, d1, l1); raw(2, d0, d1, l2); raw(3, d0, d1, l3);
            mx = fmaxf(fmaxf(l0, l1), fmaxf(l2, l3));
            mx = fmaxf(mx, __shfl_xor(mx, 1, 4));
            mx = fmaxf(mx, __shfl_xor(mx, 2, 4));
            float sum = __expf(l0 - mx) + __expf(l1 - mx) + __expf(l2 - mx) + __expf(l3 - mx);
            sum += __shfl_xor(sum, 1, 4);
            sum += __shfl_xor(sum, 2, 4);
            inv = 1.f / sum;
        }
        // pixel-space position (w_im, h_im) and attention weight of sample t (0 when the reference's
        // early-out applies or the row is padding)
        auto sample = [&](int t, float &w_im, float &h_im, float &a) {
            float oxv, oyv, lgv;
            raw(t, oxv, oyv, lgv);
            a = __expf(lgv - mx) * inv;
            float x, y;
            if (p.ref_dim == 2) {
                x = in_cur.r.x + oxv / (float)Wl;
                y = in_cur.r.y + oyv / (float)Hl;
            } else {
                x = in_cur.r.x + oxv / (float)kTP * in_cur.r.z * 0.5f;
                y = in_cur.r.y + oyv / (float)kTP * in_cur.r.w * 0.5f;
            }
            w_im = x * (float)Wl - 0.5f;
            h_im = y * (float)Hl - 0.5f;
            const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl;
            if (!inside || !active) a = 0.f;
        };

        // descriptors of my four samples
        float dw[4][4];
        uint32_t doff[4];
        uint32_t fb_mask = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float w_im, h_im, a;
            sample(t, w_im, h_im, a);
            const float ffx = floorf(w_im), ffy = floorf(h_im);
            const int x0 = (int)ffx, y0 = (int)ffy;
            const float lx = w_im - ffx, ly = h_im - ffy;
            const int tx = x0 - oxl, ty = y0 - oyl;
            const bool in_tile = tx >= 0 && ty >= 0 && tx + 1 < twl && ty + 1 < thl;
            const float hy = 1.f - ly, hx = 1.f - lx;
            const bool use_lds = in_tile && a != 0.f;
            dw[t][0] = use_lds ? hy * hx * a : 0.f;
            dw[t][1] = use_lds ? hy * lx * a : 0.f;
            dw[t][2] = use_lds ? ly * hx * a : 0.f;
            dw[t][3] = use_lds ? ly * lx * a : 0.f;
            const uint32_t r0 = use_lds ? lvl_base16 + (uint32_t)ty * pitch16 + (uint32_t)tx * 4u : lvl_base16;
            const uint32_t r1 = use_lds ? r0 + pitch16 : lvl_base16;
            doff[t] = r0 | (r1 << 16);
            // outside the staged window with a non-zero weight: published with zero weights (the LDS loop
            // stays branch-free) and accumulated afterwards by the global fallback
            if (!in_tile && a != 0.f) fb_mask |= 1u << t;
        }

        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;

#pragma unroll 1  // keeps the two rounds from being co-scheduled (256-VGPR budget at 2 workgroups/CU)
        for (int round = 0; round < 2; ++round) {
            // lanes 2*round, 2*round+1 of every quad publish their samples: slot u = (j & 1) * 4 + t
            if ((j >> 1) == round) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int u = (j & 1) * 4 + t;
                    myW[u * 16 + qd] = make_float4(dw[t][0], dw[t][1], dw[t][2], dw[t][3]);
                    myO[u * 16 + qd] = doff[t];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // two batches of 4 samples: 16 ds_read_b128 in flight each
                float4 w[4];
                uint32_t o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    w[u] = myW[(h * 4 + u) * 16 + qd];
                    o[u] = myO[(h * 4 + u) * 16 + qd];
                }
                uint4 v[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned char *a0 = tile + (o[u] & 0xffffu) * 16u + j * 16;
                    const unsigned char *a1 = tile + (o[u] >> 16) * 16u + j * 16;
                    v[u][0] = *reinterpret_cast<const uint4 *>(a0);
                    v[u][1] = *reinterpret_cast<const uint4 *>(a0 + kPixBytes);
                    v[u][2] = *reinterpret_cast<const uint4 *>(a1);
                    v[u][3] = *reinterpret_cast<const uint4 *>(a1 + kPixBytes);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    fma8(acc, v[u][0], w[u].x);
                    fma8(acc, v[u][1], w[u].y);
                    fma8(acc, v[u][2], w[u].z);
                    fma8(acc, v[u][3], w[u].w);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }

        // ---- rare: samples outside the staged windows, straight from global memory ----
        if (__any(fb_mask != 0)) {
#pragma unroll
            for (int owner = 0; owner < 4; ++owner) {
                const int W = LW[owner], H = LH[owner], start = LS[owner];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int src = (lane & ~3) | owner;
                    const int flag = __shfl((int)((fb_mask >> t) & 1u), src, 64);
                    if (!__any(flag)) continue;
                    float my_w, my_h, my_a;
                    sample(t, my_w, my_h, my_a);  // every lane recomputes ITS sample t; the owner's is broadcast
                    const float w_im = __shfl(my_w, src, 64), h_im = __shfl(my_h, src, 64);
                    const float a = __shfl(my_a, src, 64);
                    if (flag) {
                        const float ffx = floorf(w_im), ffy = floorf(h_im);
                        const int x0 = (int)ffx, y0 = (int)ffy, x1 = x0 + 1, y1 = y0 + 1;
                        const float lx = w_im - ffx, ly = h_im - ffy, hx = 1.f - lx, hy = 1.f - ly;
                        const char *lb = vbase + j * 16;
                        if (y0 >= 0 && x0 >= 0)
                            fma8(acc, *reinterpret_cast<const uint4 *>(lb + (int64_t)(start + y0 * W + x0) * kPixBytes), hy * hx * a);
                        if (y0 >= 0 && x1 <= W - 1)
                            fma8(acc, *reinterpret_cast<const uint4 *>(lb + (int64_t)(start + y0 * W + x1) * kPixBytes), hy * lx * a);
                        if (y1 <= H - 1 && x0 >= 0)
                            fma8(acc, *reinterpret_cast<const uint4 *>(lb + (int64_t)(start + y1 * W + x0) * kPixBytes), ly * hx * a);
                        if (y1 <= H - 1 && x1 <= W - 1)
                            fma8(acc, *reinterpret_cast<const uint4 *>(lb + (int64_t)(start + y1 * W + x1) * kPixBytes), ly * lx * a);
                    }
                }
            }
        }

        if (active) {
            const int64_t o = (bq * p.M + m) * kTD + j * 8;
            if (p.out_bf16) {
                *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(p.out) + o) =
                    make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]),
                               pack_bf16x2(acc[6], acc[7]));
            } else {
                float *out = reinterpret_cast<float *>(p.out) + o;
                *reinterpret_cast<float4 *>(out) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                *reinterpret_cast<float4 *>(out + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
            }
        }
        in_cur = in_nxt;
        q_cur = q_nxt;
        q_nxt = q_nn;
        i_cur += 64;
    }
}

}  // namespace sdetr

using namespace sdetr;

extern "C" void sdetr_tiled_config(int *region_w, int *region_h, int *halo)
{
    if (region_w) *region_w = kTX;
    if (region_h) *region_h = kTY;
    if (halo) *halo = kHalo;
}

extern "C" int sdetr_region_bucket(sdetr_stream_t stream, const float *ref_points, const int64_t *shapes, int ref_dim,
                                   int B, int Nq, int L, int level0_h, int level0_w, int32_t *order,
                                   int32_t *region_start, int32_t *region_box)
{
    if (B < 0 || Nq < 0 || L <= 0 || level0_h <= 0 || level0_w <= 0) return fail("region_bucket: bad dims");
    if (ref_dim != 2 && ref_dim != 4)
        return fail("Last dim of reference_points must be 2 or 4, but get %d instead.", ref_dim);
    if (!ref_points || !shapes || !order || !region_start || !region_box) return fail("region_bucket: null pointer");
    if (L != kTL) return fail("region_bucket: %d levels (the staged kernel handles exactly %d)", L, kTL);
    BucketArgs a{};
    a.ref = ref_points; a.shapes = shapes; a.q_stride = (int64_t)L * ref_dim; a.head_stride = 0; a.lvl_stride = ref_dim;
    a.n_heads = 1; a.B = B; a.Nq = Nq; a.L = L; a.H0 = level0_h; a.W0 = level0_w;
    a.RX = (level0_w + kTX - 1) / kTX;
    a.RY = (level0_h + kTY - 1) / kTY;
    a.order = order; a.region_start = region_start; a.region_box = region_box;
    const int R = a.RX * a.RY;
    if (R > 2048) return fail("region_bucket: %d regions exceed the LDS histogram", R);
    if (B == 0) return 0;
    hipLaunchKernelGGL(region_bucket_kernel, dim3((unsigned)B), dim3(1024), (size_t)(R + 16 + R * kTL * 4) * 4, stream, a);
    return check_launch("region_bucket");
}

extern "C" int sdetr_msda_tiled_forward(sdetr_stream_t stream, const void *value_hm, const int64_t *shapes,
                                        const int64_t *lsi, const float *ref, int ref_dim, const void *proj,
                                        int proj_dtype, int64_t proj_row_stride, const int32_t *order,
                                        const int32_t *region_start, const int32_t *region_box, int num_regions,
                                        int B, int Nv, int M, int D,
                                        int L, int Nq, int P, void *out, int out_dtype)
{
    if (B < 0 || Nv < 0 || M <= 0 || Nq < 0 || num_regions <= 0) return fail("msda_tiled_forward: bad dims");
    if (D != kTD || L != kTL || P != kTP)
        return fail("msda_tiled_forward: only head_dim=32, 4 levels, 4 points (got D=%d L=%d P=%d)", D, L, P);
    if (ref_dim != 2 && ref_dim != 4)
        return fail("Last dim of reference_points must be 2 or 4, but get %d instead.", ref_dim);
    if (!value_hm || !shapes || !lsi || !ref || !proj || !order || !region_start || !region_box || !out)
        return fail("msda_tiled_forward: null pointer");
    if (proj_row_stride < (int64_t)M * L * P * 3 || (proj_row_stride % 8) != 0)
        return fail("msda_tiled_forward: proj row stride must be >= 3*M*L*P and a multiple of 8");
    if ((int64_t)B * Nq == 0) return 0;
    static DeviceOnce lds_once1;
    allow_dynamic_lds(msda_tiled_kernel, lds_once1, kTiledLds);
    TiledArgs a{};
    a.value = reinterpret_cast<const char *>(value_hm); a.shapes = shapes; a.lsi = lsi; a.ref = ref; a.ref_dim = ref_dim;
    a.proj = proj; a.proj_bf16 = (proj_dtype == SDETR_BF16); a.proj_stride = proj_row_stride; a.order = order;
    a.region_start = region_start; a.region_box = region_box; a.R = num_regions; a.out = out; a.out_bf16 = (out_dtype == SDETR_BF16);
    a.B = B; a.Nv = Nv; a.M = M; a.Nq = Nq;
    const int64_t blocks = (int64_t)B * num_regions * M;
    if (blocks > 0x7fffffffLL) return fail("msda_tiled_forward: grid too large");
    hipLaunchKernelGGL(msda_tiled_kernel, dim3((unsigned)blocks), dim3(kBlock), kTiledLds, stream, a);
    note_forward_kernel(SDETR_KERNEL_MSDA_TILED);
    return check_launch("msda_tiled");
}
