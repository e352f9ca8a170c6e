// calib_pmc -- known-byte micro-launches in the access shapes of libsage_hip.so's kernels, to calibrate rocprofv3's
// FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM: only 16-B-per-lane streaming is calibrated; "other
// access widths: calibrate on a known byte count in your own access pattern").  Every kernel is launched exactly once
// per pass over buffers far larger than L2 + Infinity Cache and touches every byte it is credited with exactly once,
// so known bytes / counter value is the factor for that shape.  Prints one JSON object per kernel (known bytes, ms).
//   hipcc -O3 --offload-arch=gfx950 -o tools/calib_pmc tools/calib_pmc.hip
//   rocprofv3 --pmc FETCH_SIZE -d out -o run --output-format csv -- tools/calib_pmc
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// 16 B per lane, coalesced: the calibrated reference shape (expected FETCH_SIZE x2 = bytes)
__global__ __launch_bounds__(256) void cal_stream_read16(const uint4* __restrict__ src, size_t n16, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void cal_stream_write16(uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        dst[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
// k_dots / k_pair_cache shape: 8 lanes own a row of ROWB bytes and load it as 16-B pieces, piece = gl + 8*i; rows are a
// random permutation, each read once.  ROWB = 256: the padded D=50 float32 row (two whole 128-B lines).  ROWB = 208:
// the unpadded 13-piece layout (the lanes without a second piece re-read their first: same line, no new bytes).
template <int ROWB>
__global__ __launch_bounds__(256) void cal_gather_rows(const uint8_t* __restrict__ table, const uint32_t* __restrict__ order,
                                                       uint32_t n_rows, uint32_t* __restrict__ sink) {
    constexpr int PIECES = ROWB / 16;
    const uint32_t gl = threadIdx.x & 7u;
    uint32_t acc = 0;
    for (uint32_t r = blockIdx.x * 32u + (threadIdx.x >> 3); r < n_rows; r += gridDim.x * 32u) {
        const uint4* row = reinterpret_cast<const uint4*>(table + (size_t)order[r] * ROWB);
        const uint4 a = row[gl];
        const uint4 b = row[(8 + gl) < (uint32_t)PIECES ? 8 + gl : gl];
        acc ^= a.x ^ a.w ^ b.x ^ b.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// k_dots' store shape: 8 lanes write 8 doubles = one 64-byte chunk, chunks in random order
__global__ __launch_bounds__(256) void cal_store_chunk64(double* __restrict__ dst, const uint32_t* __restrict__ order, uint32_t n_chunks) {
    const uint32_t gl = threadIdx.x & 7u;
    for (uint32_t c = blockIdx.x * 32u + (threadIdx.x >> 3); c < n_chunks; c += gridDim.x * 32u)
        dst[(size_t)order[c] * 8 + gl] = (double)c;
}
// worst case of a key scatter: every lane stores 8 bytes to its own random slot
__global__ __launch_bounds__(256) void cal_store_scatter8(unsigned long long* __restrict__ dst, const uint32_t* __restrict__ order, uint32_t n) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) dst[order[i]] = i;
}
// the staged scatter of k_key_dedupe / k_key_partition: runs of RUN consecutive 8-byte records at random run-aligned places
template <int RUN>
__global__ __launch_bounds__(256) void cal_store_runs8(unsigned long long* __restrict__ dst, const uint32_t* __restrict__ order, uint32_t n_runs) {
    const uint32_t lanes = RUN < 64 ? RUN : 64;
    const uint32_t rl = threadIdx.x % lanes;
    for (uint32_t r = blockIdx.x * (256u / lanes) + threadIdx.x / lanes; r < n_runs; r += gridDim.x * (256u / lanes))
        for (uint32_t j = rl; j < (uint32_t)RUN; j += lanes) dst[(size_t)order[r] * RUN + j] = r;
}

static std::vector<uint32_t> permutation(uint32_t n, uint64_t seed) {
    std::vector<uint32_t> p(n);
    for (uint32_t i = 0; i < n; ++i) p[i] = i;
    uint64_t s = seed;
    for (uint32_t i = n - 1; i > 0; --i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t j = (uint32_t)((s >> 33) % (i + 1));
        const uint32_t t = p[i]; p[i] = p[j]; p[j] = t;
    }
    return p;
}

template <class F>
static void run(const char* name, double read_bytes, double write_bytes, F&& launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("{\"kernel\": \"%s\", \"known_read_bytes\": %.0f, \"known_write_bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f}\n", name,
           read_bytes, write_bytes, ms, (read_bytes + write_bytes) / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main() {
    const size_t BYTES = (size_t)2 << 30;          // 2 GiB per buffer: 8x the Infinity Cache
    uint8_t* buf = nullptr; uint8_t* buf2 = nullptr; uint32_t* order = nullptr; uint32_t* sink = nullptr;
    CK(hipMalloc(&buf, BYTES)); CK(hipMalloc(&buf2, BYTES)); CK(hipMalloc(&sink, 256));
    CK(hipMemset(buf, 1, BYTES)); CK(hipMemset(buf2, 0, BYTES));
    const uint32_t n256 = (uint32_t)(BYTES / 256), n208 = (uint32_t)(BYTES / 208), n64 = (uint32_t)(BYTES / 64);
    const uint32_t n8 = 1u << 25;                  // 32M scattered 8-byte slots over the first 256 MiB (n8 <= n_max)
    const uint32_t n_max = n64;
    CK(hipMalloc(&order, (size_t)n_max * 4));
    const uint32_t grid = 256 * 8;
    auto upload = [&](uint32_t n, uint64_t seed) {
        const std::vector<uint32_t> p = permutation(n, seed);
        CK(hipMemcpy(order, p.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    };
    run("cal_stream_read16", (double)BYTES, 0, [&] { cal_stream_read16<<<grid, 256>>>(reinterpret_cast<const uint4*>(buf), BYTES / 16, sink); });
    run("cal_stream_write16", 0, (double)BYTES, [&] { cal_stream_write16<<<grid, 256>>>(reinterpret_cast<uint4*>(buf2), BYTES / 16); });
    upload(n256, 1);
    run("cal_gather_rows<256>", (double)n256 * 256 + (double)n256 * 4, 0, [&] { cal_gather_rows<256><<<grid, 256>>>(buf, order, n256, sink); });
    upload(n208, 2);
    run("cal_gather_rows<208>", (double)n208 * 208 + (double)n208 * 4, 0, [&] { cal_gather_rows<208><<<grid, 256>>>(buf, order, n208, sink); });
    upload(n64, 3);
    run("cal_store_chunk64", (double)n64 * 4, (double)n64 * 64, [&] { cal_store_chunk64<<<grid, 256>>>(reinterpret_cast<double*>(buf2), order, n64); });
    upload(n8, 4);
    run("cal_store_scatter8", (double)n8 * 4, (double)n8 * 8, [&] { cal_store_scatter8<<<grid, 256>>>(reinterpret_cast<unsigned long long*>(buf2), order, n8); });
    const uint32_t runs8 = (uint32_t)(BYTES / 64), runs32 = (uint32_t)(BYTES / 256);
    upload(runs8, 5);
    run("cal_store_runs8<8>", (double)runs8 * 4, (double)runs8 * 64, [&] { cal_store_runs8<8><<<grid, 256>>>(reinterpret_cast<unsigned long long*>(buf2), order, runs8); });
    upload(runs32, 6);
    run("cal_store_runs8<32>", (double)runs32 * 4, (double)runs32 * 256, [&] { cal_store_runs8<32><<<grid, 256>>>(reinterpret_cast<unsigned long long*>(buf2), order, runs32); });
    CK(hipFree(buf)); CK(hipFree(buf2)); CK(hipFree(order)); CK(hipFree(sink));
    return 0;
}
