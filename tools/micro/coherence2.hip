// Which (store flavour, load flavour) pairs are coherent ACROSS XCDs inside one kernel, when the reader's
// XCD already holds the line in L2?  Writers: workgroups on XCD 0 only.  Readers: the other XCDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_fill(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 1.0f; }
__device__ void grid_barrier(int *counter, int nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblocks) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
}
// p: [0,64) plain-stored, [64,128) sc1-stored, [128,192) atomically added
__global__ void k_test(float *p, int *counter, float *out, int nblocks) {
    const int lane = threadIdx.x;
    float pre = p[lane] + p[64 + lane] + p[128 + lane];         // lines resident in this XCD's L2 / this CU's L1
    if (pre == 123.0f) out[0] = pre;
    grid_barrier(counter, nblocks);
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 0xF;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, 4096, 0x00020000);
    if (xcc == 0) {
        p[lane] = 7.0f;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, 7.0f), rs, (64 + lane) * 4, 0, 16);
        if (blockIdx.x == 0 || true) atomicAdd(p + 128 + lane, 0.0f), atomicAdd(p + 128 + lane, 0.0f);
    }
    grid_barrier(counter + 1, nblocks);
    // wait a little so that write-through traffic has landed
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(64);
    float *o = out + (size_t)blockIdx.x * 8 * 64;
    o[0 * 64 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, 0, 16));          // plain store -> sc1 load
    o[1 * 64 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (64 + lane) * 4, 0, 16));   // sc1 store  -> sc1 load
    o[2 * 64 + lane] = p[lane];                                                                                         // plain store -> plain load
    o[3 * 64 + lane] = p[64 + lane];                                                                                    // sc1 store  -> plain load
    if (lane == 0) o[7 * 64] = (float)xcc;
}
int main() {
    const int nblocks = 256;
    float *p, *out; int *counter;
    CK(hipMalloc(&p, 4096)); CK(hipMalloc(&out, sizeof(float) * nblocks * 8 * 64)); CK(hipMalloc(&counter, 8));
    CK(hipMemset(counter, 0, 8));
    k_fill<<<4, 256>>>(p, 1024);
    CK(hipDeviceSynchronize());
    k_test<<<nblocks, 64>>>(p, counter, out, nblocks);
    CK(hipDeviceSynchronize());
    std::vector<float> h(nblocks * 8 * 64);
    CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    const char *names[4] = {"plain store -> sc1 load", "sc1 store -> sc1 load", "plain store -> plain load", "sc1 store -> plain load"};
    for (int f = 0; f < 4; ++f) {
        int fresh = 0, total = 0;
        for (int b = 0; b < nblocks; ++b) { if (h[(size_t)b * 8 * 64 + 7 * 64] == 0.0f) continue;
            for (int l = 0; l < 64; ++l) { total++; if (h[(size_t)b * 8 * 64 + f * 64 + l] == 7.0f) fresh++; } }
        printf("%-28s readers on other XCDs: fresh %5d / %5d\n", names[f], fresh, total);
    }
    return 0;
}
