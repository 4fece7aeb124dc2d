// TEST INFRASTRUCTURE ONLY — a stand-in for <hip/hip_runtime.h> with exactly the host API the model-level handles use
// (csrc/model_common.*, model_layers.hip, hifigan_model.hip, vits_model.hip, glow_model.hip), so that their host logic — configuration
// checks, state_dict bookkeeping, weight folding / packing, workspace arithmetic, launch sequencing — compiles as plain C++ and runs
// on the CPU under -fsanitize=address,undefined (tests/native/handles_driver.cpp, tests/test_host_cpu.py).  "Device" memory is host
// memory of exactly the requested size (tests/native/hip_stub.cpp); graphs are not supported (the drivers pass use_graph = 0).
#pragma once
#include <cstddef>
#include <cstdint>

typedef int hipError_t;
#define hipSuccess 0
#define hipErrorOutOfMemory 2
#define hipErrorNotSupported 801
typedef struct ihipStream_t *hipStream_t;
typedef struct ihipGraph *hipGraph_t;
typedef struct hipGraphExec *hipGraphExec_t;
typedef struct ihipGraphNode *hipGraphNode_t;
typedef struct ihipEvent_t *hipEvent_t;
#define hipEventDisableTiming 2
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
#define hipStreamNonBlocking 1
#define hipHostMallocDefault 0

extern "C" {
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind kind);
hipError_t hipDeviceSynchronize(void);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode mode);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t *g);
hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t g, hipGraphNode_t *err_node, char *log, size_t log_bytes);
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s);
hipError_t hipGraphDestroy(hipGraph_t g);
hipError_t hipGraphExecDestroy(hipGraphExec_t e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipGetLastError(void);
hipError_t hipGetDevice(int *dev);
hipError_t hipFuncSetAttribute(const void *f, hipFuncAttribute a, int v);
const char *hipGetErrorString(hipError_t e);
}
