// Where the one-launch top-k (salience_detr_amd/csrc/topk.hip, topk_hsort_kernel) spends its time: the library source
// compiled with cycle stamps at its phase boundaries (workgroup 0, thread 0), on rows shaped like the two uses in the
// step -- the finest level (16 800 salience scores: a bell of ordinary scores, ~700 border near-ties at the top, 100
// masked -> fill; k = 6680) and an encoder layer's class scores (11 363, k = 300).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSDETR_HS_STAMPS -o hsort_phases benchmarks/micro/hsort_phases.hip
#include "../../salience_detr_amd/csrc/topk.hip"

#include <cmath>
#include <random>
#include <vector>

namespace sdetr {
char *error_buffer()
{
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace sdetr

// (topk.hip's launches that cannot carry a row-orders job fall back to this entry of encoder_rows.hip; no case here has a job)
extern "C" int sdetr_layer_row_orders(sdetr_stream_t, const int64_t *, int64_t, const int32_t *, int, int, int, int, const int32_t *,
                                      int32_t *, int64_t)
{
    return 0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int run(const char *name, int B, int n, int k, int near_ties, int masked, bool last)
{
    std::mt19937 rng(1234);
    std::normal_distribution<float> bell(-0.42f, 0.16f);
    std::vector<float> h((size_t)B * n);
    std::vector<uint8_t> hm((size_t)B * n, 0);
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < n; ++i) h[(size_t)b * n + i] = bell(rng);
        for (int i = 0; i < near_ties; ++i) {   // spread over the row like image borders; values a few ulps apart
            const int pos = (int)(((int64_t)i * 7919) % n);
            h[(size_t)b * n + pos] = std::nextafter(0.1985638f, 1.f + (float)(i % 5));
            for (int u = 0; u < i % 5; ++u) h[(size_t)b * n + pos] = std::nextafter(h[(size_t)b * n + pos], 1.f);
        }
        for (int i = 0; i < masked; ++i) hm[(size_t)b * n + n - 1 - i * 3] = 1;
    }
    float *score, *out_score, *fillv;
    uint8_t *mask;
    int64_t *out_index;
    void *ws;
    CK(hipMalloc(&score, h.size() * 4));
    CK(hipMalloc(&mask, hm.size()));
    CK(hipMalloc(&out_score, (size_t)B * k * 4));
    CK(hipMalloc(&out_index, (size_t)B * k * 8));
    CK(hipMalloc(&fillv, 16));
    CK(hipMalloc(&ws, 1 << 20));
    const float fv = -1.5f;
    CK(hipMemcpy(score, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(mask, hm.data(), hm.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(fillv, &fv, 4, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto call = [&]() {
        return sdetr_masked_topk_desc_f32(s, score, masked ? mask : nullptr, n, masked ? 2 : 0, masked ? fillv : nullptr, nullptr, B, n, k,
                                          0, out_score, out_index, k, ws, 1 << 20);
    };
    for (int r = 0; r < 3; ++r)
        if (call()) { fprintf(stderr, "%s\n", sdetr::error_buffer()); return 1; }
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 20; ++r) call();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long st[16];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(sdetr::hs_stamps), sizeof(st)));
    // s_memtime counts at 100 MHz on this chip (a constant clock, not the shader clock)
    printf(" {\"case\": \"%s\", \"n\": %d, \"k\": %d, \"us_per_launch\": %.2f, \"phase_ticks\": {\"load_reduce\": %llu, "
           "\"histogram\": %llu, \"scan\": %llu, \"scatter\": %llu, \"rank_ordinary\": %llu, \"crowded\": %llu, \"floor\": %llu}}%s\n",
           name, n, k, ms * 1e3f / 20, st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4],
           st[6] - st[5], st[7] - st[6], last ? "" : ",");
    return 0;
}

int main()
{
    printf("{\"hsort_phases\": [\n");
    if (run("level0", 2, 16800, 6680, 726, 100, false)) return 1;
    if (run("level0_no_near_ties", 2, 16800, 6680, 0, 100, false)) return 1;
    if (run("layer0_top300", 2, 11363, 300, 0, 0, false)) return 1;
    if (run("layer5_top300", 2, 2272, 300, 0, 0, true)) return 1;
    printf("]}\n");
    return 0;
}
