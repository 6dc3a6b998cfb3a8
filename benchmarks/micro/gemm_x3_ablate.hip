// Where csrc/gemm_x3.hip spends its time: the library source compiled with one leg removed (SDETR_GX3_ABLATE, see there),
// timed on the feed-forward's first product (22 726 x 256 -> 2048).  Build one binary per variant:
//   for v in 0 1 2 3; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DSDETR_GX3_ABLATE=$v \
//       -I include -o gemm_x3_ablate_$v benchmarks/micro/gemm_x3_ablate.hip; done
#include "../../salience_detr_amd/csrc/gemm_x3.hip"

#include <vector>

namespace sdetr {
char *error_buffer()
{
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace sdetr

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 22726, K = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 2048;
    const bool dw = argc > 4 && argv[4][0] == 'd';   // "dw": the weight gradient dy^T x (both operands reduction-major, 16 slices)
    if (argc > 5) sdetr_gemm_x3_generation(atoi(argv[5]));   // 1 = 128 x 128 tiles, 2 = 256 x 128 tiles, default: the shape rule
    float *x, *w, *y;
    CK(hipMalloc(&x, (size_t)T * K * 4));
    CK(hipMalloc(&w, (size_t)N * K * 4));
    CK(hipMalloc(&y, (size_t)T * N * 4));
    std::vector<float> h((size_t)T * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8) * (1.f / 16777216.f) - 0.5f;
    CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    h.resize((size_t)N * K);
    CK(hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // y = x w^T [T, N]   or   dw = dy^T x [N, K] with dy = the [T, N] buffer (reduction over T in 16 slices, atomics)
    auto call = [&]() {
        if (dw) return sdetr_gemm_x3_f32(s, y, N, 0, x, K, 0, w, K, N, K, T, nullptr, 16, nullptr);
        return sdetr_gemm_x3_f32(s, x, K, 1, w, K, 1, y, N, T, N, K, nullptr, 1, nullptr);
    };
    for (int r = 0; r < 3; ++r)
        if (call()) { fprintf(stderr, "%s\n", sdetr::error_buffer()); return 1; }
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 10; ++r) call();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"ablate\": %d, \"product\": \"%s\", \"T\": %d, \"K\": %d, \"N\": %d, \"us\": %.1f, \"tflops_fp32_equivalent\": %.1f}\n", SDETR_GX3_ABLATE,
           dw ? "dw = dy^T x" : "y = x w^T", T, K, N,
           ms * 100.f, 2.0 * T * K * N / (ms * 1e-4) / 1e12);
    return 0;
}
