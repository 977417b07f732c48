// Where do the seconds of a 100 GB device allocation go, and does the virtual-memory API (physical chunks created in parallel,
// mapped into one reserved range) get there faster?   hipcc -O2 -o alloc_probe alloc_probe.cpp -lpthread ; ./alloc_probe [GB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(double *p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i * 512] = 1.0; }
int main(int argc, char **argv)
{
    const size_t GB = argc > 1 ? atoll(argv[1]) : 100;
    const size_t bytes = GB << 30;
    CK(hipSetDevice(0));
    CK(hipFree(0));
    double t0, t1, t2;
    void *p = nullptr;
    auto dirty = [&]() -> int {   // leave `bytes` of freed, written memory behind (what a previous tile store looks like)
        void *q = nullptr;
        CK(hipMalloc(&q, bytes));
        CK(hipMemset(q, 0x5a, bytes));
        CK(hipDeviceSynchronize());
        CK(hipFree(q));
        return 0;
    };
    t0 = now(); CK(hipMalloc(&p, bytes)); t1 = now(); CK(hipFree(p));
    printf("first hipMalloc %zu GB of the process: %.3f s\n", GB, t1 - t0);
    for (int rep = 0; rep < 2; rep++) {
        if (dirty()) return 1;
        t0 = now(); CK(hipMalloc(&p, bytes)); t1 = now();
        hipLaunchKernelGGL(touch, dim3((unsigned)((bytes / 4096 + 255) / 256)), dim3(256), 0, 0, (double *)p, bytes / 4096);
        CK(hipGetLastError()); CK(hipDeviceSynchronize());
        t2 = now(); CK(hipFree(p));
        printf("hipMalloc after a freed store: %.3f s (touch %.3f s)\n", t1 - t0, t2 - t1);
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran < (2u << 20)) gran = 2u << 20;
    for (int nchunk : {8, 8, 32}) {
        for (int threaded = 0; threaded < 2; threaded++) {
            if (dirty()) return 1;
            size_t chunk = ((bytes / nchunk + gran - 1) / gran) * gran;
            void *va = nullptr;
            t0 = now();
            CK(hipMemAddressReserve(&va, chunk * nchunk, 0, nullptr, 0));
            std::vector<hipMemGenericAllocationHandle_t> h(nchunk);
            std::vector<int> rc(nchunk, 0);
            auto make = [&](int i) { (void)hipSetDevice(0); rc[i] = (int)hipMemCreate(&h[i], chunk, &prop, 0); };
            if (threaded) { std::vector<std::thread> th; for (int i = 0; i < nchunk; i++) th.emplace_back(make, i); for (auto &t : th) t.join(); }
            else for (int i = 0; i < nchunk; i++) make(i);
            for (int i = 0; i < nchunk; i++) if (rc[i]) { printf("hipMemCreate chunk %d failed: %d\n", i, rc[i]); return 1; }
            t1 = now();
            for (int i = 0; i < nchunk; i++) CK(hipMemMap((char *)va + (size_t)i * chunk, chunk, 0, h[i], 0));
            hipMemAccessDesc ad = {};
            ad.location.type = hipMemLocationTypeDevice; ad.location.id = 0; ad.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, chunk * nchunk, &ad, 1));
            t2 = now();
            hipLaunchKernelGGL(touch, dim3((unsigned)((bytes / 4096 + 255) / 256)), dim3(256), 0, 0, (double *)va, bytes / 4096);
            CK(hipGetLastError()); CK(hipDeviceSynchronize());
            double back = 0.0;
            CK(hipMemcpy(&back, (char *)va + (bytes / 4096 - 1) * 4096, 8, hipMemcpyDeviceToHost));
            double t3b = now();
            CK(hipMemUnmap(va, chunk * nchunk));
            for (int i = 0; i < nchunk; i++) CK(hipMemRelease(h[i]));
            CK(hipMemAddressFree(va, chunk * nchunk));
            double t4 = now();
            printf("VMM %2d chunks %s after a freed store: create %.3f s, map+access %.3f s, touch %.3f s (read back %.1f), teardown %.3f s\n",
                   nchunk, threaded ? "threaded  " : "sequential", t1 - t0, t2 - t1, t3b - t2, back, t4 - t3b);
        }
    }
    return 0;
}
