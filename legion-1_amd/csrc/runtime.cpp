// runtime.cpp -- error handling, logical->physical device map and the raw device helpers
// (reference: src/Kernels.cu:14-64, Kernels.cuh:14-45).
#include "internal.h"

#include <cstring>
#include <mutex>

#include "audit_hooks.h"

namespace legion {

static int g_error_mode = LEGION_ERR_EXIT;
static thread_local std::string t_last_error;
static int g_dev_map[64];
static bool g_dev_map_set[64];
static std::mutex g_mu;
static bool g_remote[64];

void report_error(const char* file, int line, const char* msg, bool hip_failure)
{
    char buf[1024];
    // reference text: "Cuda failure %s:%d: '%s'" (Kernels.cuh:18)
    snprintf(buf, sizeof(buf), "%s failure %s:%d: '%s'", hip_failure ? "Hip" : "Legion", file, line, msg);
    if (hip_failure && g_error_mode == LEGION_ERR_EXIT) {
        fprintf(log_file(), "%s\n", buf);
        fflush(log_file());
        exit(EXIT_FAILURE);
    }
    if (t_last_error.empty()) t_last_error = buf; // sticky: first error wins
    if (!hip_failure && g_error_mode == LEGION_ERR_EXIT) fprintf(stderr, "%s\n", buf);
    (void)hipGetLastError(); // clear HIP's own sticky flag so later calls are attributable
}

static int log_to_stderr()
{
    static const int v = [] { const char* e = getenv("LEGION_LOG"); return (e && strcmp(e, "stderr") == 0) ? 1 : 0; }();
    return v;
}
std::ostream& log_out() { return log_to_stderr() ? std::cerr : std::cout; }
FILE* log_file() { return log_to_stderr() ? stderr : stdout; }

bool error_pending() { return !t_last_error.empty(); }
bool error_is_fatal() { return g_error_mode == LEGION_ERR_EXIT; }

// Largest single allocation that may cross a process boundary as a HIP IPC handle: see LEGION_IPC_MAX_BYTES_DEFAULT in
// include/legion_amd.h (the HIP runtime bundled with the torch wheel, ROCm 7.0, hangs in hipIpcOpenMemHandle at >= 2^31
// bytes; profiles/r02_ipc_limit.md).
int64_t ipc_max_bytes()
{
    const char* e = getenv("LEGION_IPC_MAX_BYTES");
    return e && atoll(e) > 0 ? atoll(e) : LEGION_IPC_MAX_BYTES_DEFAULT;
}
bool ipc_size_ok(int64_t bytes, const char* who)
{
    if (bytes <= ipc_max_bytes()) return true;
    char msg[320];
    snprintf(msg, sizeof(msg), "%s: a single allocation of %lld bytes exceeds the HIP-IPC limit of %lld bytes (LEGION_IPC_MAX_BYTES): "
             "larger single allocations never finish importing on this driver; use the chunked calls / a smaller buffer",
             who, (long long)bytes, (long long)ipc_max_bytes());
    report_error(__FILE__, __LINE__, msg, false);
    return false;
}
bool ipc_export_ok(const void* ptr, const char* who)
{
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr) != hipSuccess) { (void)hipGetLastError(); return true; }
    return ipc_size_ok((int64_t)size, who);
}

bool is_remote_device(int logical) { return logical >= 0 && logical < 64 && g_remote[logical]; }

int physical_device(int logical)
{
    if (logical >= 0 && logical < 64 && g_dev_map_set[logical]) return g_dev_map[logical];
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    return logical >= 0 ? logical % n : 0;
}

DeviceGuard::DeviceGuard(int logical)
{
    HIP_CHECK(hipGetDevice(&prev));
    int want = physical_device(logical);
    if (want != prev) HIP_CHECK(hipSetDevice(want));
    else prev = -1;
    prev_logical = current_logical_device();
    set_current_logical_device(logical);
}
DeviceGuard::~DeviceGuard()
{
    if (prev >= 0) (void)hipSetDevice(prev);
    set_current_logical_device(prev_logical);
}

} // namespace legion

using namespace legion;

extern "C" {

const char* legion_version(void) { return "legion-amd 0.1.0 (gfx950)"; }
int32_t legion_row_pitch(int32_t F)
{
    if (F <= 0 || (F * 4) % 128 == 0) return F;
    return (F + 31) / 32 * 32;
}
void legion_set_error_mode(int mode) { g_error_mode = mode; }
const char* legion_last_error(void) { return t_last_error.c_str(); }
void legion_clear_error(void) { t_last_error.clear(); }

void legion_set_device_map(int32_t logical_dev, int32_t physical_dev)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (logical_dev < 0 || logical_dev >= 64) { LEGION_ARG_ERROR("legion_set_device_map: logical id out of range"); return; }
    g_dev_map[logical_dev] = physical_dev;
    g_dev_map_set[logical_dev] = true;
}
int32_t legion_physical_device(int32_t logical_dev) { return physical_device(logical_dev); }
void legion_set_remote_device(int32_t logical_dev, int is_remote)
{
    if (logical_dev < 0 || logical_dev >= 64) { LEGION_ARG_ERROR("legion_set_remote_device: logical id out of range"); return; }
    g_remote[logical_dev] = is_remote != 0;
}
int legion_is_remote_device(int32_t logical_dev) { return is_remote_device(logical_dev) ? 1 : 0; }

// ---- Kernels.cu:14-64 ---------------------------------------------------------------------------
void* d_alloc_space(int64_t num_bytes)
{
    void* ret = nullptr;
    HIP_CHECK(hipMalloc(&ret, (size_t)(num_bytes > 0 ? num_bytes : 1)));
    return ret;
}
void* d_alloc_space_managed(unsigned int num_bytes)
{
    void* ret = nullptr;
    HIP_CHECK(hipMallocManaged(&ret, num_bytes ? num_bytes : 1));
    return ret;
}
void d_copy_2_h(void* h_ptr, void* d_ptr, unsigned int num_bytes)
{
    HIP_CHECK(hipMemcpy(h_ptr, d_ptr, num_bytes, hipMemcpyDeviceToHost));
}
void d_free_space(void* d_ptr) { (void)hipFree(d_ptr); }
void SetGPUDevice(int32_t shard_id)
{
    HIP_CHECK(hipSetDevice(physical_device(shard_id)));
    set_current_logical_device(shard_id);
}
int32_t GetGPUDevice(void)
{
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    return dev;
}
void* host_alloc_space64(int64_t num_bytes)
{
    void* host_ptr = nullptr;
    void* ret = nullptr;
    HIP_CHECK(hipHostMalloc(&host_ptr, (size_t)(num_bytes > 0 ? num_bytes : 1), hipHostMallocMapped | hipHostMallocPortable));
    if (!host_ptr) return nullptr;
    HIP_CHECK(hipHostGetDevicePointer(&ret, host_ptr, 0));
    return ret;
}
void* host_alloc_space(unsigned int num_bytes) { return host_alloc_space64((int64_t)num_bytes); }
void host_free_space(void* ptr) { (void)hipHostFree(ptr); }
void d_copy_h_2_d(void* d_ptr, const void* h_ptr, int64_t num_bytes)
{
    HIP_CHECK(hipMemcpy(d_ptr, h_ptr, (size_t)num_bytes, hipMemcpyHostToDevice));
}
void d_copy_d_2_h(void* h_ptr, const void* d_ptr, int64_t num_bytes)
{
    HIP_CHECK(hipMemcpy(h_ptr, d_ptr, (size_t)num_bytes, hipMemcpyDeviceToHost));
}
void d_stream_sync(void* stream) { HIP_CHECK(hipStreamSynchronize((hipStream_t)stream)); }
void* d_stream_create(void)
{
    hipStream_t s = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return (void*)s;
}
void* d_stream_create_priority(int32_t high)
{   // high != 0: greatest priority of the device's range, else the least
    hipStream_t s = nullptr;
    int least = 0, greatest = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high ? greatest : least));
    return (void*)s;
}
void* d_stream_create_cu_mask(const uint32_t* cu_mask, int32_t words)
{   // a stream whose kernels only run on the CUs whose bit is set (hipExtStreamCreateWithCUMask)
    hipStream_t s = nullptr;
    if (!cu_mask || words <= 0) { LEGION_ARG_ERROR("d_stream_create_cu_mask: empty mask"); return nullptr; }
    HIP_CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, cu_mask));
    return (void*)s;
}
void d_stream_destroy(void* stream) { (void)hipStreamDestroy((hipStream_t)stream); }
void d_copy_async(void* dst, const void* src, int64_t num_bytes, void* stream)
{
    HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)num_bytes, hipMemcpyDefault, (hipStream_t)stream));
}
void d_memset_async(void* dst, int value, int64_t num_bytes, void* stream)
{
    HIP_CHECK(hipMemsetAsync(dst, value, (size_t)num_bytes, (hipStream_t)stream));
}
void* d_event_create(void)
{
    hipEvent_t e = nullptr;
    HIP_CHECK(hipEventCreate(&e));
    return (void*)e;
}
void d_event_destroy(void* event) { (void)hipEventDestroy((hipEvent_t)event); }
void d_event_record(void* event, void* stream) { HIP_CHECK(hipEventRecord((hipEvent_t)event, (hipStream_t)stream)); }
void d_stream_wait_event(void* stream, void* event) { HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0)); }
float d_event_elapsed_ms(void* start, void* stop)
{
    float ms = 0.f;
    HIP_CHECK(hipEventSynchronize((hipEvent_t)stop));
    HIP_CHECK(hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop));
    return ms;
}

void legion_rng_probe(void* stream, const int32_t* idx, const int32_t* deg, int32_t* k_out, int32_t n)
{
    launch_rng_probe((hipStream_t)stream, idx, deg, k_out, n);
}

} // extern "C"
