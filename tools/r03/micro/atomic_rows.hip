// atomic_rows.hip -- what would it cost to APPLY a streamed row's update with float atomic adds instead of storing the row?
// (the Hogwild lag is two workers' read-modify-write of the same ordinary row inside a few microseconds; an atomic add of
// the delta cannot lose the other worker's update.)  5 120 wavefronts visit random 1 664-byte records of a 2 GB table, as the trainer
// does: read the record (sc1), then write it back in one of four ways, timed with HIP events.
//   0  dwordx4 load, dwordx4 store                         -- what the trainer does today
//   1  dwordx4 load, four strided dword atomic adds        -- the lane keeps its four floats; every atomic touches 4 of each 16 bytes
//   2  dword loads of consecutive floats, dword stores     -- the record transposed: lane l holds floats l, l+64, ...
//   3  dword loads of consecutive floats, dword atomic adds -- seven atomics of 256 contiguous bytes each
//   hipcc --offload-arch=gfx950 -O2 -o atomic_rows atomic_rows.hip ;  ./atomic_rows [records in the table]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef int i4 __attribute__((ext_vector_type(4)));
constexpr int SC1 = 16;
constexpr int REC_FLOATS = 416, REC_BYTES = REC_FLOATS * 4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_rows(float *tab, int64_t rows, int32_t iters, uint32_t seed) {
    const int lane = threadIdx.x & 63;
    uint32_t x = (uint32_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 0x9E3779B1u + seed;
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        const int64_t r = (int64_t)(((uint64_t)(x >> 4) * (uint64_t)rows) >> 28);
        const __amdgpu_buffer_rsrc_t rs = rsrc(tab + r * REC_FLOATS, REC_BYTES);
        if constexpr (MODE == 0 || MODE == 1) {
            i4 v[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (lane + q * 64) * 16, 0, SC1);
            if constexpr (MODE == 0) {
#pragma unroll
                for (int q = 0; q < 2; ++q) __builtin_amdgcn_raw_buffer_store_b128(v[q], rs, (lane + q * 64) * 16, 0, SC1);
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int c = 0; c < 4; ++c)          // delta = 1e-30 x what was read: the table stays where it is
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(__builtin_bit_cast(float, v[q][c]) * 1e-30f, rs, (lane + q * 64) * 16 + c * 4, 0, 0);
            }
        } else {
            int v[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b32(rs, (lane + q * 64) * 4, 0, SC1);
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                if constexpr (MODE == 2) __builtin_amdgcn_raw_buffer_store_b32(v[q], rs, (lane + q * 64) * 4, 0, SC1);
                else __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(__builtin_bit_cast(float, v[q]) * 1e-30f, rs, (lane + q * 64) * 4, 0, 0);
            }
        }
    }
}

int main(int argc, char **argv) {
    const int64_t rows = argc > 1 ? std::atoll(argv[1]) : 1250000;     // 1 250 000 records = 2 GB; 60 000 = 100 MB: inside the Infinity Cache
    const int iters = 4000;                       // 5 120 wavefronts x 4 000 records = 20.5 M record visits (an epoch of the bench: 104 M)
    float *tab = nullptr;
    CHECK(hipMalloc((void **)&tab, rows * REC_BYTES));
    CHECK(hipMemset(tab, 0, rows * REC_BYTES));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    void (*k[4])(float *, int64_t, int32_t, uint32_t) = {k_rows<0>, k_rows<1>, k_rows<2>, k_rows<3>};
    const char *what[4] = {"dwordx4 load + dwordx4 store", "dwordx4 load + 4 strided atomics", "dword loads + dword stores", "dword loads + 7 contiguous atomics"};
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 4; ++m) {
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(k[m], dim3(1280), dim3(256), 0, 0, tab, rows, iters, 12345u + rep);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
            const double visits = 5120.0 * iters;
            std::printf("{\"rows\": %lld, \"mode\": %d, \"what\": \"%s\", \"rep\": %d, \"ms\": %.3f, \"record_visits_per_s\": %.4g, \"GBps_read_plus_write\": %.1f}\n",
                        (long long)rows, m, what[m], rep, ms, visits / ms * 1e3, visits * 2 * REC_BYTES / ms * 1e-6);
        }
    return 0;
}
