// TEST INFRASTRUCTURE ONLY — the host-memory "device" behind tests/native/hip_stub/hip/hip_runtime.h: every hipMalloc is a malloc of
// exactly the requested size (the address sanitizer then sees any access past a workspace / weight image), hipMemcpy a memcpy,
// streams are tokens, graphs are unsupported.  g_hip_stub_fail_after >= 0 makes the n-th following hipMalloc fail (error paths).
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>

long g_hip_stub_fail_after = -1;
long g_hip_stub_live = 0;

extern "C" {
hipError_t hipMalloc(void **p, size_t n)
{
    if (g_hip_stub_fail_after == 0) {
        g_hip_stub_fail_after = -1;
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    if (g_hip_stub_fail_after > 0) --g_hip_stub_fail_after;
    *p = malloc(n ? n : 1);
    if (*p) ++g_hip_stub_live;
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p)
{
    if (p) --g_hip_stub_live;
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void *p) { return hipFree(p); }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind)
{
    memcpy(dst, src, n);
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned)
{
    *s = reinterpret_cast<hipStream_t>(0x10);
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g)
{
    *g = nullptr;
    return hipErrorNotSupported;
}
hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t, hipGraphNode_t *, char *, size_t)
{
    *e = nullptr;
    return hipErrorNotSupported;
}
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned)
{
    *e = static_cast<hipEvent_t>(malloc(1));        // a real allocation: an event that is never destroyed shows up as a leak
    return *e ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e)
{
    free(e);
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipGetDevice(int *dev)
{
    *dev = 0;
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : (e == hipErrorOutOfMemory ? "hipErrorOutOfMemory (stub)" : "hip error (stub)"); }
}
