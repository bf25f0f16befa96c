// xcd_locality.hip -- does the distance between an XCD and the memory behind a page show, and at what granularity?
// One wave per XCD walks a sample of 4 KiB pages of a 1 GiB buffer with dependent loads (sc1, lines nobody touched before: XCD x
// uses lines 8x .. 8x+7 of each page) and records the mean load-to-use time per (XCD, page).  Run for a hipMalloc'ed buffer and for a
// physically contiguous one (hipExtMallocWithFlags + hipDeviceMallocContiguous).  Output: one CSV row per page.
//   hipcc --offload-arch=gfx950 -O2 -o xcd_locality xcd_locality.hip ;  ./xcd_locality [contiguous] > out.csv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int PAGES = 7936;          // sampled pages: 7936 x 132 KiB stays inside the GiB
constexpr int LINES = 8;             // dependent loads per (XCD, page)

__global__ __launch_bounds__(64) void k_probe(const char *buf, size_t page_stride, int *claimed, unsigned *out /* [8][PAGES] */) {
    // XCC_ID: hardware register 20, bits 3:0 (gfx940+)
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;
    __shared__ int mine;
    if (threadIdx.x == 0) mine = atomicAdd(&claimed[xcc], 1) == 0;      // the first workgroup that lands on this XCD does the work
    __syncthreads();
    if (!mine || threadIdx.x != 0) return;
    unsigned acc = 0;
    for (int p = 0; p < PAGES; ++p) {
        const char *page = buf + (size_t)p * page_stride;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        unsigned off = xcc * LINES * 64;
        for (int k = 0; k < LINES; ++k) {
            // the next address depends on the loaded value (the buffer holds zeros): a true dependent chain
            const unsigned v = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(page + off + (acc & 1u)));
            acc += v;
            off += 64 + (v & 1u);
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        out[xcc * PAGES + p] = (unsigned)((t1 - t0) / LINES) + (acc & 1u);
    }
}

int main(int argc, char **argv) {
    const bool contiguous = argc > 1 && std::strcmp(argv[1], "contiguous") == 0;
    const size_t bytes = (size_t)1 << 30;
    char *buf = nullptr;
    if (contiguous) CHECK(hipExtMallocWithFlags((void **)&buf, bytes, hipDeviceMallocContiguous));
    else CHECK(hipMalloc((void **)&buf, bytes));
    CHECK(hipMemset(buf, 0, bytes));
    int *claimed = nullptr; unsigned *out = nullptr;
    CHECK(hipMalloc((void **)&claimed, 8 * sizeof(int)));
    CHECK(hipMalloc((void **)&out, 8 * PAGES * sizeof(unsigned)));
    CHECK(hipMemset(claimed, 0, 8 * sizeof(int)));
    CHECK(hipMemset(out, 0, 8 * PAGES * sizeof(unsigned)));
    CHECK(hipDeviceSynchronize());
    // pages 132 KiB + 4 KiB apart: walks through every 4 KiB slot of 2 MiB blocks while covering the whole GiB
    const size_t page_stride = 135168;
    hipLaunchKernelGGL(k_probe, dim3(2048), dim3(64), 0, 0, buf, page_stride, claimed, out);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> h(8 * PAGES);
    CHECK(hipMemcpy(h.data(), out, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    std::printf("# %s buffer at %p, page stride %zu bytes, mean ticks per dependent load\npage,offset", contiguous ? "contiguous" : "hipMalloc", (void *)buf, page_stride);
    for (int x = 0; x < 8; ++x) std::printf(",xcd%d", x);
    std::printf("\n");
    for (int p = 0; p < PAGES; ++p) {
        std::printf("%d,%zu", p, (size_t)p * page_stride);
        for (int x = 0; x < 8; ++x) std::printf(",%u", h[(size_t)x * PAGES + p]);
        std::printf("\n");
    }
    return 0;
}
