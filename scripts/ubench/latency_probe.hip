// Memory-path latencies of THIS box, one wave, dependent chains (why do some boxes / processes run the
// single-sentence loop 25 % slower at the same shader clock?): pointer chase through global_load (64-bit addresses), through
// buffer_load (resource + 32-bit offset), through scalar loads, over a footprint that sits in L2 (256 KB) and one that does not
// (256 MB); an empty-kernel launch chain; reported in ns per step.
//   hipcc --offload-arch=gfx950 -O3 latency_probe.hip -o latency_probe && ./latency_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void chase_global(const unsigned *p, unsigned start, int steps, unsigned *out, long long *cyc)
{
    unsigned i = start;
    const long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) i = p[i];
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { *out = i; *cyc = t1 - t0; }
}
__global__ void chase_buffer(const unsigned *p, unsigned bytes, unsigned start, int steps, unsigned *out, long long *cyc)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(p), 0, (int)bytes, 0x00020000);
    unsigned i = start;
    const long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) i = __builtin_amdgcn_raw_buffer_load_b32(r, (int)(i * 4u), 0, 0);
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { *out = i; *cyc = t1 - t0; }
}
__global__ void chase_scalar(const unsigned *__restrict__ p, unsigned start, int steps, unsigned *out, long long *cyc)
{
    unsigned i = __builtin_amdgcn_readfirstlane(start);          // wave-uniform index: the compiler walks the chain with s_load_dword
    const long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) i = __builtin_amdgcn_readfirstlane(p[i]);
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { *out = i; *cyc = t1 - t0; }
}
__global__ void empty_kernel() {}

int main()
{
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    int wc_khz = 0;
    CK(hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0));
    printf("device %s  CUs %d  clock %d MHz  mem clock %d MHz  wall clock %d kHz  L2 %d KB  pci %04x:%02x:%02x  gcnArch %s\n", pr.name, pr.multiProcessorCount,
           pr.clockRate / 1000, pr.memoryClockRate / 1000, wc_khz, pr.l2CacheSize / 1024, pr.pciDomainID, pr.pciBusID, pr.pciDeviceID, pr.gcnArchName);
    unsigned *out;
    long long *cyc;
    CK(hipMalloc(&out, 4));
    CK(hipMalloc(&cyc, 8));
    for (size_t words : {size_t(64) << 10, size_t(64) << 20}) {             // 256 KB (L2) and 256 MB (HBM)
        std::vector<unsigned> h(words);
        // one cycle through all cache lines (stride 64 words = 256 B, scrambled) so that every step is a fresh line
        const size_t lines = words / 64;
        std::vector<unsigned> order(lines);
        for (size_t i = 0; i < lines; ++i) order[i] = (unsigned)i;
        unsigned long long x = 88172645463325252ull;
        for (size_t i = lines - 1; i > 0; --i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            std::swap(order[i], order[x % (i + 1)]);
        }
        for (size_t i = 0; i < lines; ++i) h[(size_t)order[i] * 64] = order[(i + 1) % lines] * 64;
        unsigned *d;
        CK(hipMalloc(&d, words * 4));
        CK(hipMemcpy(d, h.data(), words * 4, hipMemcpyHostToDevice));
        const int steps = 4096;
        unsigned start = 0;                      // every run continues the chain where the last one stopped: lines not touched before
        for (int kind = 0; kind < 3; ++kind) {
            double best = 1e30;
            for (int rep = 0; rep < 5; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(chase_global, dim3(1), dim3(64), 0, 0, d, start, steps, out, cyc);
                else if (kind == 1) hipLaunchKernelGGL(chase_buffer, dim3(1), dim3(64), 0, 0, d, (unsigned)(words * 4 > 0x7fffffffull ? 0x7fffffff : words * 4), start, steps, out, cyc);
                else hipLaunchKernelGGL(chase_scalar, dim3(1), dim3(64), 0, 0, d, start, steps, out, cyc);
                CK(hipDeviceSynchronize());
                long long c;
                CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(&start, out, 4, hipMemcpyDeviceToHost));
                const double ns = (double)c / (wc_khz * 1e-6) / steps;       // wall clock ticks -> ns
                if (ns < best) best = ns;
            }
            printf("  %-7s footprint %4zu %s: %-22s %7.1f ns per dependent load\n", words * 4 >= (1u << 20) * 100 ? "HBM" : "L2", words * 4 >> (words * 4 >= (1 << 20) ? 20 : 10),
                   words * 4 >= (1 << 20) ? "MB" : "KB", kind == 0 ? "global_load (64-bit)" : kind == 1 ? "buffer_load (rsrc)" : "scalar load", best);
        }
        CK(hipFree(d));
    }
    // launch chain: 2000 empty kernels back to back on one stream
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0);
        CK(hipDeviceSynchronize());
        auto t1 = std::chrono::steady_clock::now();
        if (rep == 2) printf("  empty-kernel chain: %.2f us per launch (2000 launches, one stream)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000);
    }
    return 0;
}
