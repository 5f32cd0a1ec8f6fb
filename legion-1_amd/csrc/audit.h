// audit.h -- logical-device audit of the library's own HIP traffic ($LEGION_DEVICE_AUDIT=1).
//
// Why: the reference runs one host thread per GPU, each under cudaSetDevice(own) (Server.cu:87-95,119-127), with cache shards and
// CSR fragments allocated on their owner and read by the clique's other GPUs through peer access enabled all-pairs at boot
// (GPUGraphStore.cu:145-168, GPU_Memory_Graph_Storage.cu:98-133, GPUCache.cu:769-826).  This library addresses GPUs by LOGICAL id
// (legion_set_device_map) so that a clique can be exercised on one physical device -- but there a stream, event, allocation or launch
// made under the wrong device works all the same (DeviceGuard compares physical ids), and fails, or silently puts a shard on the
// wrong GPU, on the first real node.  The audit makes such mistakes visible on ONE GPU:
//   * a thread-local LOGICAL current device, maintained by SetGPUDevice / DeviceGuard (always, audit or not);
//   * every stream, event, device / pinned / managed allocation, VMM region, IPC import and instantiated graph the library creates is
//     tagged with the logical device it was created under (audit_hooks.h routes the HIP calls of every source file here);
//   * every kernel launch (LEGION_AUDIT_LAUNCH in the launch wrappers), event record, stream wait, asynchronous copy / memset and graph
//     launch checks stream tag == current logical device; HIP's own current device must be the physical device of the logical one;
//   * a kernel argument the kernel WRITES, or reads as its own scratch, must be memory of the current logical device (nothing in this
//     library writes into a peer); a TABLE it reads (CSR, feature rows, rankings, hotness) may also be pinned host memory, an IPC
//     import, or memory of another logical device if -- and only if -- peer access between the two logical devices was recorded by the
//     all-pairs enable (storage.cpp enable_p2p); the same rule for the pointers inside device-side pointer tables (shard / fragment
//     chunk tables); both ends of a hipMemcpyPeerAsync must live on the physical devices the call names;
//   * places that fill a per-device slot state the owner they expect (audit::expect_owner).
// A violation is a sticky error naming the call site, is kept in a process-wide list (legion_audit_*) and counted; the `legion`
// server prints the summary when it stops and exits non-zero on a violation.  Off (the default): one predictable branch per call.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <initializer_list>

namespace legion {

int current_logical_device();                 // logical GPU selected on this thread (SetGPUDevice / DeviceGuard); -1: none yet
void set_current_logical_device(int logical);

namespace audit {

extern bool g_on;
inline bool on() { return g_on; }

struct Arg { const void* p; char mode; const char* name; };   // mode 'w': written, 'l': read, scratch of this GPU (both: must be local) | 'r': read, may be a peer's

void launch(hipStream_t s, const char* kernel, const char* file, int line, std::initializer_list<Arg> args);
// the pointers of a device-side pointer table that logical GPU `viewer` dereferences in-kernel
void table(int viewer, const void* const* ptrs, size_t n, const char* what, const char* file, int line);
// `p` (null: ignored) must be memory tagged with logical GPU `logical`
void expect_owner(const void* p, int logical, const char* what, const char* file, int line);
// a stream / event must carry the tag `logical`
void expect_stream(hipStream_t s, int logical, const char* what, const char* file, int line);
// one allocation that serves several logical GPUs because they share a physical device (a replica per physical device)
void share(const void* p, int logical);
void record_peer(int a, int b);               // peer access a -> b was enabled
void region(void* va, size_t bytes, int logical, const char* file, int line);   // a mapped VMM range
void region_gone(void* va);

// ---- the HIP calls of the library, routed here by audit_hooks.h ----
hipError_t Malloc(void** p, size_t n, const char* f, int l);
hipError_t MallocManaged(void** p, size_t n, const char* f, int l);
hipError_t Free(void* p, const char* f, int l);
hipError_t HostMalloc(void** p, size_t n, unsigned flags, const char* f, int l);
hipError_t HostFree(void* p, const char* f, int l);
hipError_t IpcOpen(void** p, hipIpcMemHandle_t h, unsigned flags, const char* f, int l);
hipError_t IpcClose(void* p, const char* f, int l);
hipError_t StreamCreateWithFlags(hipStream_t* s, unsigned flags, const char* f, int l);
hipError_t StreamCreateWithPriority(hipStream_t* s, unsigned flags, int prio, const char* f, int l);
hipError_t StreamCreateWithCUMask(hipStream_t* s, uint32_t words, const uint32_t* mask, const char* f, int l);
hipError_t StreamDestroy(hipStream_t s, const char* f, int l);
hipError_t EventCreate(hipEvent_t* e, const char* f, int l);
hipError_t EventCreateWithFlags(hipEvent_t* e, unsigned flags, const char* f, int l);
hipError_t EventDestroy(hipEvent_t e, const char* f, int l);
hipError_t EventRecord(hipEvent_t e, hipStream_t s, const char* f, int l);
hipError_t StreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags, const char* f, int l);
hipError_t Memcpy(void* dst, const void* src, size_t n, hipMemcpyKind kind, const char* f, int l);
hipError_t MemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s, const char* f, int l);
hipError_t Memcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t w, size_t h, hipMemcpyKind kind, const char* f, int l);
hipError_t MemcpyPeerAsync(void* dst, int ddev, const void* src, int sdev, size_t n, hipStream_t s, const char* f, int l);
hipError_t Memset(void* dst, int v, size_t n, const char* f, int l);
hipError_t MemsetAsync(void* dst, int v, size_t n, hipStream_t s, const char* f, int l);
hipError_t GraphInstantiate(hipGraphExec_t* exec, hipGraph_t g, hipGraphNode_t* en, char* log, size_t n, const char* f, int l);
hipError_t GraphLaunch(hipGraphExec_t exec, hipStream_t s, const char* f, int l);
hipError_t GraphExecDestroy(hipGraphExec_t exec, const char* f, int l);
hipError_t StreamBeginCapture(hipStream_t s, hipStreamCaptureMode mode, const char* f, int l);

} // namespace audit
} // namespace legion

#define LEGION_AW(p) ::legion::audit::Arg{(const void*)(p), 'w', #p}
#define LEGION_AL(p) ::legion::audit::Arg{(const void*)(p), 'l', #p}
#define LEGION_AR(p) ::legion::audit::Arg{(const void*)(p), 'r', #p}
#define LEGION_AUDIT_LAUNCH(s, kernel, ...) \
    do { if (::legion::audit::on()) ::legion::audit::launch((s), (kernel), __FILE__, __LINE__, {__VA_ARGS__}); } while (0)
#define LEGION_AUDIT_TABLE(viewer, ptrs, n, what) \
    do { if (::legion::audit::on()) ::legion::audit::table((viewer), (const void* const*)(ptrs), (n), (what), __FILE__, __LINE__); } while (0)
#define LEGION_AUDIT_OWNER(p, logical, what) \
    do { if (::legion::audit::on()) ::legion::audit::expect_owner((const void*)(p), (logical), (what), __FILE__, __LINE__); } while (0)
#define LEGION_AUDIT_STREAM(s, logical, what) \
    do { if (::legion::audit::on()) ::legion::audit::expect_stream((hipStream_t)(s), (logical), (what), __FILE__, __LINE__); } while (0)
#define LEGION_AUDIT_SHARE(p, logical) \
    do { if (::legion::audit::on()) ::legion::audit::share((const void*)(p), (logical)); } while (0)
