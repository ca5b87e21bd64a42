// Test double of the HIP runtime for the host-side sanitizer builds (tests/hostcheck): just the calls abi.hip and ntru_host.hip
// make, with "device" memory on the heap and every stream a worker thread that runs enqueued work IN ORDER and ASYNCHRONOUSLY --
// so ThreadSanitizer sees a host write into a staging buffer that an enqueued copy has not read yet, and AddressSanitizer sees a
// copy that runs past an arena.  Not a GPU emulation: the kernels are replaced by the test doubles of fake_device.cpp.
#ifndef NTRU_FAKE_HIP_RUNTIME_H
#define NTRU_FAKE_HIP_RUNTIME_H
#include <cstddef>
#include <cstdint>
#include <functional>

typedef enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 } hipError_t;
struct FakeStream;
struct FakeEvent;
typedef FakeStream *hipStream_t;
typedef FakeEvent *hipEvent_t;
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; };
typedef enum { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeUnregistered = 0 } hipMemoryType;
struct hipPointerAttribute_t { hipMemoryType type; };
typedef enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;
typedef enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 } hipFuncAttribute;
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDisableTiming = 2, hipHostRegisterDefault = 0 };

const char *hipGetErrorString(hipError_t e);
hipError_t hipGetLastError(void);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipMalloc(void **p, size_t bytes);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipHostRegister(void *p, size_t bytes, unsigned flags);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *at, const void *p);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipFuncSetAttribute(const void *fn, hipFuncAttribute a, int v);
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *fn, int threads, size_t lds);

// test-double "kernel launch": run `work` on the stream's worker thread, after everything enqueued before it
void fake_enqueue(hipStream_t s, std::function<void()> work);
#endif
