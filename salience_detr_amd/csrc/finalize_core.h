// The token-space pass of the encoder's output as a job other launches carry (fused_head_value.hip: a filtering launch;
// topk.hip: the merge of the finest level's sliced top-k).
#pragma once
#include "common.h"

namespace sdetr {

// out[b, s, :] = tokens[b, s, :] + (pad[b, s] ? 0 : background[s, :]), bf16 rows of 256 -- the token-space pass of the
// encoder's output (sdetr_encoder_finalize's first launch): it depends on nothing the filtering or the encoder compute,
// so a filtering launch carries it; the sorted rows are overwritten at the very end as before.
struct FinalizeJob {
    const uint4 *tokens;       // [B * S * 32] 16-byte pieces
    const uint4 *background;   // [S * 32]
    const uint8_t *pad;        // [B * S] or NULL
    uint4 *out;
    int64_t total;             // B * S * 32
    int S;
};
// Four pieces per thread and trip: all their loads (token piece, padding byte, background piece) are requested before the
// first sum -- one piece per trip was a chain of ~14 dependent round trips per thread (23 us as a rider of 192 blocks).
__device__ __forceinline__ void finalize_all_role(const FinalizeJob &j, int role_block, int role_blocks)
{
    const int64_t stride = (int64_t)role_blocks * blockDim.x;
    for (int64_t t0 = (int64_t)role_block * blockDim.x + threadIdx.x; t0 < j.total; t0 += 4 * stride) {
        uint4 a[4], g[4];
        bool live[4], padded[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t t = t0 + u * stride;
            live[u] = t < j.total;
            const int64_t tc = live[u] ? t : j.total - 1;
            const int64_t r = tc >> 5;   // b * S + s
            a[u] = j.tokens[tc];
            padded[u] = j.pad && j.pad[r];
            // (r < 2^31 is checked where the job is built: a 32-bit modulo instead of a 64-bit one per piece; the
            // background piece is read whether the token is padding or not -- no load behind a loaded condition)
            g[u] = j.background[(int64_t)((uint32_t)r % (uint32_t)j.S) * 32 + (int)(tc & 31)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!live[u]) continue;
            uint4 o = a[u];
            if (!padded[u])
                o = make_uint4(pack_act2(act_lo(a[u].x) + act_lo(g[u].x), act_hi(a[u].x) + act_hi(g[u].x)),
                               pack_act2(act_lo(a[u].y) + act_lo(g[u].y), act_hi(a[u].y) + act_hi(g[u].y)),
                               pack_act2(act_lo(a[u].z) + act_lo(g[u].z), act_hi(a[u].z) + act_hi(g[u].z)),
                               pack_act2(act_lo(a[u].w) + act_lo(g[u].w), act_hi(a[u].w) + act_hi(g[u].w)));
            j.out[t0 + u * stride] = o;
        }
    }
}

// sdetr_finalize_job -> FinalizeJob (checked); returns 0 or the error code of fail()
static inline int fill_finalize_job(FinalizeJob &fj, const sdetr_finalize_job *finalize, const char *who)
{
    if (finalize->batch <= 0 || finalize->spatial_size <= 0 || !finalize->tokens || !finalize->background || !finalize->out)
        return fail("%s: bad finalize job", who);
    if ((int64_t)finalize->batch * finalize->spatial_size >= ((int64_t)1 << 31))
        return fail("%s: finalize job too large for 32-bit token arithmetic", who);
    fj.tokens = (const uint4 *)finalize->tokens; fj.background = (const uint4 *)finalize->background;
    fj.pad = finalize->padding_mask; fj.out = (uint4 *)finalize->out; fj.S = finalize->spatial_size;
    fj.total = (int64_t)finalize->batch * finalize->spatial_size * 32;
    return 0;
}

}  // namespace sdetr
