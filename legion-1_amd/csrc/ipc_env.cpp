// ipc_env.cpp -- the server <-> trainer hand-off.
//   server half : CUDAIPCEnv            src/CUDA_IPC_Service.cu:39-360
//   trainer half: GPUIPCEnv             pytorch_extension/ipc_cuda_kernel.cu:36-176
//   shm helpers : helper_multiprocess   src/helper_multiprocess.cpp:5-100
// Contract kept byte for byte: POSIX shm "simpleIPCshm" holding
//   struct { int32 steps[3]; ipcMemHandle[8 dev][2 pipe][7 buf]; }   (12 + 8*2*7*64 = 7180 B;
// hipIpcMemHandle_t is 64 bytes like cudaIpcMemHandle_t), named semaphores sem_r_D_P / sem_w_D_P
// with initial value 0, producer: wait(sem_r) -> fill -> post(sem_w); consumer: wait(sem_w) ->
// use -> post(sem_r).  Extensions (hop count, counter mirror, feature-buffer rows) live in a SECOND shm
// object "<name>_ext" (shmExt below): the shared slab itself is exactly the reference's 7180 bytes.
#include "internal.h"

#include <cerrno>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <iostream>
#include <algorithm>
#include <optional>
#include <atomic>
#include <vector>
#include <poll.h>
#include <semaphore.h>
#include <sys/mman.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <thread>
#include <unistd.h>

#include "audit_hooks.h"

using namespace legion;

static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle must be 64 bytes (CUDA_IPC_Service.cu:34-37)");

struct shmStruct {           // the reference's slab, byte for byte (CUDA_IPC_Service.cu:34-37, ipc_cuda_kernel.cu:31-34): 7180 bytes, nothing appended
    int32_t steps[3];
    hipIpcMemHandle_t memHandle[LEGION_MAX_DEVICE][LEGION_PIPELINE_DEPTH][LEGION_MEMORY_USAGE];
};
static_assert(sizeof(shmStruct) == 12 + 8 * 2 * 7 * 64, "the reference's shm layout is 7180 bytes");
// Everything this implementation adds lives in a SECOND shm object, "<name>_ext".  Rounds 1-4 appended it behind the reference struct in the
// same object; but the reference's trainer opens the slab with sharedMemoryCreate (ipc_cuda_kernel.cu:45 -> helper_multiprocess.cpp:35), which
// ftruncate()s it to ITS 7180 bytes -- cutting the appended words off under a running server (a SIGBUS on their next touch beyond the page).
// With a separate object the shared one is exactly the reference's, whoever creates or truncates it (tests/test_ref_shm_compat.py drives the
// reference's own helper against it).  A peer without the extension object simply does not find it: hops = 2, counters by device copy.
struct shmExt {
    uint32_t mirror_magic;    // kMirrorMagic: the server maintains the counter mirror below
    int32_t hops;             // number of hops the server samples (0: not told -> 2, the reference's layout)
    // rows the feature buffers of a device hold (InitializeFeaturesBuffer; 0 = not told).  The reference sizes them at 1.2 x the largest batch
    // of the pre-sampling epoch (Server.cu:275) and its trainer views [nc9, F] of them unchecked (ipc_cuda_kernel.cu:200): a batch with more
    // nodes reads past the allocation.  A client that knows the capacity refuses such a batch instead.
    int32_t feature_rows[LEGION_MAX_DEVICE];
    // host mirror of the two 16-int counter arrays of every (device, pipe), filled by the server before it posts the pipe.  The reference's
    // trainer reads them with a blocking cudaMemcpy from the IPC device buffers (ipc_cuda_kernel.cu:195-196): on the legacy default stream that
    // copy waits for everything the trainer has queued.  A client that finds the magic set reads the mirror instead; buffers 5 / 6 stay valid.
    int32_t counters[LEGION_MAX_DEVICE][LEGION_PIPELINE_DEPTH][32];   // nc[16] | ec[16]
    // What ties this object to the LIVE slab: the object is only unlinked by IPCEnv_Finalize, so after a killed server a server WITHOUT the
    // extension (the reference's) would leave a stale one in place and a client would read stale hops / counters for good.  The server stores
    // a copy of the slab's step counts and a checksum of the first handle it registers per device; a client ignores an extension that does not
    // match the slab it attached to (it then behaves as against a reference server).
    int32_t steps_copy[3];
    uint32_t handle_sum[LEGION_MAX_DEVICE];   // FNV-1a of memHandle[dev][0][0]; 0 = nothing registered for that device yet
};
static const uint32_t kMirrorMagic = 0x4C474E43u;   // "LGNC"
static uint32_t handle_checksum(const volatile void* h)
{
    uint32_t x = 2166136261u;
    const volatile unsigned char* b = (const volatile unsigned char*)h;
    for (size_t i = 0; i < sizeof(hipIpcMemHandle_t); i++) { x ^= b[i]; x *= 16777619u; }
    return x ? x : 1u;
}

// $LEGION_IPC_NO_DEVICE=1 (test hook: build containers without a GPU, the host-sanitizer run of tests/test_ipc_env_cpu.py): the slab,
// the named semaphores, the counter mirror and the poisoned-pipe protocol with no device call at all -- no hand-off buffer is
// allocated, the handle slots stay zero (a client takes an all-zero handle as "no buffer"), nothing is page-locked.
static bool no_device()
{
    static const bool v = [] { const char* e = getenv("LEGION_IPC_NO_DEVICE"); return e && e[0] == '1'; }();
    return v;
}
#define LEGION_DEVICE_GUARD(dev) std::optional<DeviceGuard> guard_; if (!no_device()) guard_.emplace(dev)

// Waiting for the other side of the hand-off: poll the semaphore for a bounded time before blocking on it.  A blocked waiter is woken
// through the kernel (futex) -- 10-60 us on an idle core -- and that latency sits on the depth-2 handshake of every batch: the server may only
// refill a pipe after the trainer has handed it back.  A waiter polls for 200 us first (profiles/r05_handoff_spin.log: 0 / 50 / 200 / 1000 us
// measured, 200 kept).  A trainer that is the bottleneck costs the server at most that much polling per batch, then it sleeps as before.
static int handoff_spin_us() { return 200; }
static inline int64_t mono_ns()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}
// sem_wait with a polling prologue; EINTR-safe
static void sem_wait_spin(sem_t* sem)
{
    const int spin = handoff_spin_us();
    if (spin > 0) {
        const int64_t until = mono_ns() + (int64_t)spin * 1000;
        do {
            if (sem_trywait(sem) == 0) return;
            for (int i = 0; i < 32; i++) __builtin_ia32_pause();
        } while (mono_ns() < until);
    }
    while (sem_wait(sem) != 0 && errno == EINTR) {}
}

static std::string g_namespace;
static bool g_ns_init = false;
static const std::string& ipc_ns()
{
    if (!g_ns_init) {
        const char* e = getenv("LEGION_IPC_NAMESPACE");
        if (e) g_namespace = e;
        g_ns_init = true;
    }
    return g_namespace;
}
static std::string shm_name() { return "/" + ipc_ns() + "simpleIPCshm"; }
static std::string ext_name() { return shm_name() + "_ext"; }
static std::string sem_name(const char* rw, int dev, int pipe)
{
    return "/" + ipc_ns() + "sem_" + rw + "_" + std::to_string(dev) + "_" + std::to_string(pipe);
}

// sharedMemoryCreate (helper_multiprocess.cpp:5-47): shm_open(O_RDWR|O_CREAT) + ftruncate + mmap
static void* shm_map(size_t sz, int* fd_out, const std::string& name = shm_name(), bool create = true)
{
    int fd = shm_open(name.c_str(), create ? (O_RDWR | O_CREAT) : O_RDWR, 0777);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) == 0 && (size_t)st.st_size < sz && ftruncate(fd, (off_t)sz) != 0) { close(fd); return nullptr; }
    void* addr = mmap(nullptr, sz, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (addr == MAP_FAILED) { close(fd); return nullptr; }
    *fd_out = fd;
    return addr;
}

// ---- hand-off buffers above the HIP-IPC size limit -------------------------------------------------------------------
// hipIpcOpenMemHandle of the runtime bundled with the torch wheel (ROCm 7.0) never returns for an allocation of 2^31 bytes or
// more (profiles/r02_ipc_limit.md), and a trainer needs its feature rows as ONE contiguous tensor.  Such a buffer is therefore
// built from <= 1 GiB physical chunks (HIP virtual memory management), exported as POSIX file descriptors and handed to the
// trainer over an abstract unix socket (SCM_RIGHTS); the trainer maps the chunks back to back into one reserved virtual range
// (profiles/vmm_ipc_probe.*: 5 GiB mapped in 21 ms under ROCm 7.2, 3 GiB in 0.3 ms inside a PyTorch process).  The 64-byte
// handle slot of the shm table carries a descriptor instead of an IPC handle.
struct VmmDesc {               // lives in shmStruct::memHandle[dev][pipe][1]
    char magic[8];             // "LGNVMM01"
    uint64_t total, chunk;     // bytes mapped, bytes per chunk (the last one may be shorter)
    uint32_t nchunks, pad;
};
static_assert(sizeof(VmmDesc) <= sizeof(hipIpcMemHandle_t), "descriptor must fit the handle slot");
static const char kVmmMagic[8] = {'L', 'G', 'N', 'V', 'M', 'M', '0', '1'};

static socklen_t vmm_sock_addr(sockaddr_un* a, int dev, int pipe)
{
    memset(a, 0, sizeof(*a));
    a->sun_family = AF_UNIX;
    const std::string name = ipc_ns() + "legion_vmm_" + std::to_string(dev) + "_" + std::to_string(pipe);
    const size_t n = std::min(name.size(), sizeof(a->sun_path) - 2);
    memcpy(a->sun_path + 1, name.data(), n);       // abstract namespace: sun_path[0] == 0, nothing to unlink
    return (socklen_t)(offsetof(sockaddr_un, sun_path) + 1 + n);
}
static bool send_fd(int sock, int fd)
{
    char byte = 'f';
    iovec io{&byte, 1};
    alignas(cmsghdr) char ctrl[CMSG_SPACE(sizeof(int))] = {0};
    msghdr msg{};
    msg.msg_iov = &io; msg.msg_iovlen = 1; msg.msg_control = ctrl; msg.msg_controllen = sizeof(ctrl);
    cmsghdr* c = CMSG_FIRSTHDR(&msg);
    c->cmsg_level = SOL_SOCKET; c->cmsg_type = SCM_RIGHTS; c->cmsg_len = CMSG_LEN(sizeof(int));
    memcpy(CMSG_DATA(c), &fd, sizeof(int));
    return sendmsg(sock, &msg, MSG_NOSIGNAL) == 1;
}
static int recv_fd(int sock)
{
    char byte;
    iovec io{&byte, 1};
    alignas(cmsghdr) char ctrl[CMSG_SPACE(sizeof(int))] = {0};
    msghdr msg{};
    msg.msg_iov = &io; msg.msg_iovlen = 1; msg.msg_control = ctrl; msg.msg_controllen = sizeof(ctrl);
    if (recvmsg(sock, &msg, 0) != 1) return -1;
    cmsghdr* c = CMSG_FIRSTHDR(&msg);
    int fd = -1;
    if (c && c->cmsg_level == SOL_SOCKET && c->cmsg_type == SCM_RIGHTS) memcpy(&fd, CMSG_DATA(c), sizeof(int));
    return fd;
}

struct VmmRegion {             // server side: one mapped, exported buffer + the thread that serves its descriptors
    void* va = nullptr;
    VmmDesc desc{};
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<int> fds;
    int listen_fd = -1, device = 0;
    std::thread server;
    std::atomic<bool> stop{false};
};
static void vmm_serve(VmmRegion* r)
{
    while (!r->stop.load()) {
        pollfd p{r->listen_fd, POLLIN, 0};
        if (poll(&p, 1, 200) <= 0) continue;
        const int s = accept4(r->listen_fd, nullptr, nullptr, SOCK_CLOEXEC);
        if (s < 0) continue;
        // the descriptors are read/write dmabuf handles of the feature buffer: only hand them to a process of this user,
        // and never let a stalled client hold the serving thread (IPCEnv_Finalize joins it)
        timeval tv{2, 0};
        (void)setsockopt(s, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
        (void)setsockopt(s, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
        ucred cred{};
        socklen_t cl = sizeof(cred);
        bool ok = getsockopt(s, SOL_SOCKET, SO_PEERCRED, &cred, &cl) == 0 && cred.uid == geteuid();
        ok = ok && send(s, &r->desc, sizeof(r->desc), MSG_NOSIGNAL) == (ssize_t)sizeof(r->desc);
        for (size_t i = 0; ok && i < r->fds.size(); i++) ok = send_fd(s, r->fds[i]);
        // the trainer acknowledges after it has mapped everything; wait for it in slices so that a shutdown is never blocked
        for (int waited = 0; ok && waited < 30 && !r->stop.load(); waited++) {
            char ack;
            const ssize_t n = recv(s, &ack, 1, 0);
            if (n >= 0 || (errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR)) break;
        }
        close(s);
    }
}
static void vmm_release(VmmRegion* r)
{
    r->stop.store(true);
    if (r->server.joinable()) r->server.join();
    if (r->listen_fd >= 0) close(r->listen_fd);
    for (int fd : r->fds) close(fd);
    if (r->va) {
        if (audit::on()) audit::region_gone(r->va);
        (void)hipMemUnmap(r->va, r->desc.total);
        for (auto h : r->handles) (void)hipMemRelease(h);
        (void)hipMemAddressFree(r->va, r->desc.total);
    }
    delete r;
}
// nullptr (sticky error) when the platform cannot do it
static VmmRegion* vmm_create(int logical_dev, int pipe, size_t bytes)
{
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = physical_device(logical_dev);
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || !gran) {
        (void)hipGetLastError();
        LEGION_ARG_ERROR("hand-off buffer above the HIP-IPC limit: this runtime has no virtual memory management");
        return nullptr;
    }
    const char* e = getenv("LEGION_SHARD_CHUNK_BYTES");     // one chunk size for everything that crosses a process boundary in pieces
    size_t chunk = e && atoll(e) > 0 ? (size_t)atoll(e) : ((size_t)1 << 30);
    chunk = std::max(gran, chunk / gran * gran);
    VmmRegion* r = new VmmRegion();
    r->device = logical_dev;
    const size_t total = (bytes + gran - 1) / gran * gran;
    memcpy(r->desc.magic, kVmmMagic, 8);
    r->desc.total = total; r->desc.chunk = chunk; r->desc.nchunks = (uint32_t)((total + chunk - 1) / chunk);
    bool ok = hipMemAddressReserve(&r->va, total, 0, nullptr, 0) == hipSuccess;
    for (uint32_t c = 0; ok && c < r->desc.nchunks; c++) {
        const size_t sz = std::min(chunk, total - (size_t)c * chunk);
        hipMemGenericAllocationHandle_t h;
        ok = hipMemCreate(&h, sz, &prop, 0) == hipSuccess;
        if (!ok) break;
        r->handles.push_back(h);
        ok = hipMemMap((char*)r->va + (size_t)c * chunk, sz, 0, h, 0) == hipSuccess;
        int fd = -1;
        ok = ok && hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0) == hipSuccess;
        if (ok) r->fds.push_back(fd);
    }
    hipMemAccessDesc acc{};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    ok = ok && hipMemSetAccess(r->va, total, &acc, 1) == hipSuccess;
    if (ok) {
        sockaddr_un addr;
        const socklen_t len = vmm_sock_addr(&addr, logical_dev, pipe);
        r->listen_fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
        ok = r->listen_fd >= 0 && bind(r->listen_fd, (sockaddr*)&addr, len) == 0 && listen(r->listen_fd, 8) == 0;
    }
    if (!ok) {
        (void)hipGetLastError();
        LEGION_ARG_ERROR("hand-off buffer above the HIP-IPC limit: building the chunked (virtual memory) buffer failed");
        r->desc.total = r->va ? total : 0;
        vmm_release(r);
        return nullptr;
    }
    if (audit::on()) audit::region(r->va, total, logical_dev, __FILE__, __LINE__);
    r->server = std::thread(vmm_serve, r);
    return r;
}
// osHandle: the ROCm 7.0 runtime (the one bundled with the torch wheel, hipRuntimeGetVersion 70051831) dereferences it as
// int*, ROCm 7.2 (70226015) takes the descriptor by value (like CUDA).  The convention is picked ONCE from the version of the
// runtime this process really loaded -- no trial call: a genuine failure of the first form (stale fd, out of memory) must come
// back as an error, not be retried in the form that makes the other runtime dereference a small integer.
static hipError_t vmm_import_fd(hipMemGenericAllocationHandle_t* h, int fd)
{
    static int by_pointer = -1;
    if (by_pointer < 0) {
        int v = 0;
        if (hipRuntimeGetVersion(&v) != hipSuccess) { (void)hipGetLastError(); v = 0; }
        by_pointer = v < 70100000;
    }
    static thread_local int fd_cell;
    fd_cell = fd;
    return hipMemImportFromShareableHandle(h, by_pointer ? (void*)&fd_cell : (void*)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor);
}

struct IPCEnv {
    volatile shmStruct* shm = nullptr;
    int shm_fd = -1;
    volatile shmExt* ext = nullptr;    // the extension object ("<name>_ext"); never null once NewIPCEnv succeeded
    int ext_fd = -1;
    int32_t device_count = 0;
    std::vector<std::vector<void*>> ids, float_features, labels, agg_src, agg_dst, node_counter, edge_counter;
    std::vector<std::vector<sem_t*>> semr, semw;
    int32_t raw_batch_size = 0;
    std::vector<int32_t> train_batch_size, valid_batch_size, test_batch_size;
    int32_t train_step = 0, valid_step = 0, test_step = 0, epoch = 0, pipeline_depth = LEGION_PIPELINE_DEPTH;
    std::vector<VmmRegion*> vmm;   // feature buffers above the HIP-IPC size limit (chunked, mapped, served over a unix socket)
    bool shm_pinned = false;       // the slab is registered with the runtime: asynchronous copies can target the counter mirror
    bool mirror_fresh[LEGION_MAX_DEVICE][LEGION_PIPELINE_DEPTH] = {};   // IPCEnv_MirrorCounters ran for the batch about to be posted
    int32_t* mirror_stage[LEGION_MAX_DEVICE][LEGION_PIPELINE_DEPTH] = {};  // slab not page-locked: queued copies land in pinned staging words
};

extern "C" {

void legion_ipc_set_namespace(const char* ns)
{
    g_namespace = ns ? ns : "";
    g_ns_init = true;
}
// Remove what a server of namespace `ns` that was KILLED left in /dev/shm (IPCEnv_Finalize never ran): the slab, its extension object and
// the 2 x depth named semaphores of each of `devices` GPUs.  Safe to call when nothing is there.
void legion_ipc_unlink_namespace(const char* ns, int32_t devices)
{
    const std::string p = ns ? ns : "";
    shm_unlink(("/" + p + "simpleIPCshm").c_str());
    shm_unlink(("/" + p + "simpleIPCshm_ext").c_str());
    for (int d = 0; d < devices && d < LEGION_MAX_DEVICE; d++)
        for (int q = 0; q < LEGION_PIPELINE_DEPTH; q++)
            for (const char* rw : {"r", "w"}) sem_unlink(("/" + p + "sem_" + rw + "_" + std::to_string(d) + "_" + std::to_string(q)).c_str());
}

static void pin_slab(IPCEnv* e);

// ============================ server half ==============================================================
// CUDAIPCEnv::CUDAIPCEnv, CUDA_IPC_Service.cu:41-65
IPCEnv* NewIPCEnv(int32_t device_count)
{
    if (device_count < 1 || device_count > LEGION_MAX_DEVICE) { LEGION_ARG_ERROR("NewIPCEnv: device_count must be 1..8"); return nullptr; }
    IPCEnv* e = new IPCEnv();
    log_out() << "start initialize ipc env\n";
    e->shm = (volatile shmStruct*)shm_map(sizeof(shmStruct), &e->shm_fd);
    if (!e->shm) {
        fprintf(log_file(), "Failed to create shared memory slab\n"); // reference: exit(EXIT_FAILURE) (CUDA_IPC_Service.cu:46-48)
        delete e;
        LEGION_ARG_ERROR("NewIPCEnv: shm_open/mmap failed");
        return nullptr;
    }
    e->ext = (volatile shmExt*)shm_map(sizeof(shmExt), &e->ext_fd, ext_name());
    if (!e->ext) {
        munmap((void*)e->shm, sizeof(shmStruct)); close(e->shm_fd);
        delete e;
        LEGION_ARG_ERROR("NewIPCEnv: shm_open/mmap of the extension object failed");
        return nullptr;
    }
    log_out() << "Shared Memory Opened\n";
    memset((void*)e->shm, 0, sizeof(shmStruct));
    memset((void*)e->ext, 0, sizeof(shmExt));
    e->device_count = device_count;
    auto rs = [&](std::vector<std::vector<void*>>& v) { v.assign(device_count, {}); };
    rs(e->ids); rs(e->float_features); rs(e->labels); rs(e->agg_src); rs(e->agg_dst); rs(e->node_counter); rs(e->edge_counter);
    e->semr.assign(device_count, {});
    e->semw.assign(device_count, {});
    pin_slab(e);
    return e;
}

// Page-lock the slab so that a copy engine can write the counter mirror (hipMemcpyAsync into pageable memory is staged and
// synchronous).  Called with a device current; failure only costs the asynchronous path.
static void pin_slab(IPCEnv* e)
{
    if (e->shm_pinned || !e->ext) return;
    const size_t page = (size_t)sysconf(_SC_PAGESIZE), bytes = (sizeof(shmExt) + page - 1) / page * page;
    const char* no_pin = getenv("LEGION_IPC_NO_PIN");      // test hook: behave as if the runtime refused (the staging path below)
    if (no_device()) { e->ext->mirror_magic = kMirrorMagic; return; }
    if (!(no_pin && no_pin[0] == '1') && hipHostRegister((void*)e->ext, bytes, hipHostRegisterPortable) == hipSuccess) e->shm_pinned = true;
    else (void)hipGetLastError();
    e->ext->mirror_magic = kMirrorMagic;    // the mirror is maintained either way (synchronously in IPCPost if need be)
}

// Coordinate, CUDA_IPC_Service.cu:66-134
void IPCEnv_Coordinate(IPCEnv* e, const LegionBuildInfo* info)
{
    if (!e || !info) { LEGION_ARG_ERROR("IPCEnv_Coordinate: null argument"); return; }
    if (info->partition_count < 1 || info->partition_count > e->device_count || info->raw_batch_size < 1 || info->epoch < 0 ||
        !info->training_set_num || !info->validation_set_num || !info->testing_set_num) {
        // the reference divides by raw_batch_size unchecked (CUDA_IPC_Service.cu:89)
        LEGION_ARG_ERROR("IPCEnv_Coordinate: partition_count must be 1..device_count, raw_batch_size >= 1, epoch >= 0, the three size arrays non-null");
        return;
    }
    const int32_t P = info->partition_count;
    e->epoch = info->epoch;
    e->raw_batch_size = info->raw_batch_size;
    int32_t min_train_size = 1000000000;
    for (int32_t i = 0; i < P; i++) min_train_size = std::min(min_train_size, info->training_set_num[i]);
    e->train_step = (min_train_size - 1) / e->raw_batch_size;
    e->train_batch_size.assign(P, e->raw_batch_size);
    int32_t max_valid_size = 0, max_test_size = 0;
    const int32_t raw_valid_batch_size = 512, raw_test_batch_size = 512;
    for (int32_t i = 0; i < P; i++) max_valid_size = std::max(max_valid_size, info->validation_set_num[i]);
    e->valid_step = (max_valid_size - 1) / raw_valid_batch_size + 1;
    e->valid_batch_size.resize(P);
    for (int32_t i = 0; i < P; i++) e->valid_batch_size[i] = (info->validation_set_num[i] - 1) / e->valid_step + 1;
    for (int32_t i = 0; i < P; i++) max_test_size = std::max(max_test_size, info->testing_set_num[i]);
    e->test_step = (max_test_size - 1) / raw_test_batch_size + 1;
    e->test_batch_size.resize(P);
    for (int32_t i = 0; i < P; i++) e->test_batch_size[i] = (info->testing_set_num[i] - 1) / e->test_step + 1;
    log_out() << "Train Steps: " << e->train_step << "\n";
    log_out() << "Valid Steps: " << e->valid_step << "\n";
    log_out() << "Test Steps: " << e->test_step << "\n";
    e->shm->steps[0] = e->train_step;
    e->shm->steps[1] = e->valid_step;
    e->shm->steps[2] = e->test_step;
    if (e->ext) for (int i = 0; i < 3; i++) e->ext->steps_copy[i] = e->shm->steps[i];
}

int32_t IPCEnv_GetMaxStep(IPCEnv* e) { return ((e->train_step + e->valid_step) * e->epoch) + e->test_step; }

static void* ipc_alloc(volatile shmStruct* shm, int dev, int pipe, int which, size_t bytes)
{
    void* p = nullptr;
    if (no_device()) return nullptr;      // handle slot stays zero
    // a trainer would stall forever inside ipc_service.initialize() on a buffer it cannot import
    if (!ipc_size_ok((int64_t)bytes, which == 1 ? "InitializeFeaturesBuffer (rows x F x 4 bytes of one pipe)" : "InitializeSamplesBuffer")) {
        if (!error_is_fatal()) return nullptr;
        fflush(stderr);
        exit(EXIT_FAILURE);
    }
    HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16));
    LEGION_AUDIT_OWNER(p, dev, "hand-off buffer");
    if (p) HIP_CHECK(hipIpcGetMemHandle((hipIpcMemHandle_t*)&shm->memHandle[dev][pipe][which], p));
    return p;
}

// InitializeSamplesBuffer, CUDA_IPC_Service.cu:140-201
void IPCEnv_InitializeSamplesBuffer(IPCEnv* e, int32_t batch_size, int32_t num_ids, int32_t feature_dim,
                                    int32_t device_id, int32_t pipeline_depth)
{
    (void)feature_dim;
    if (!e || device_id < 0 || device_id >= e->device_count || pipeline_depth < 1 || pipeline_depth > LEGION_PIPELINE_DEPTH) { LEGION_ARG_ERROR("InitializeSamplesBuffer: bad arguments"); return; }
    LEGION_DEVICE_GUARD(device_id);
    e->semr[device_id].assign(pipeline_depth, nullptr);
    e->semw[device_id].assign(pipeline_depth, nullptr);
    for (int32_t i = 0; i < pipeline_depth; i++) {
        e->ids[device_id].push_back(ipc_alloc(e->shm, device_id, i, 0, (size_t)num_ids * sizeof(int32_t)));
        e->labels[device_id].push_back(ipc_alloc(e->shm, device_id, i, 2, (size_t)batch_size * sizeof(int32_t)));
        e->agg_src[device_id].push_back(ipc_alloc(e->shm, device_id, i, 3, (size_t)num_ids * sizeof(int32_t)));
        e->agg_dst[device_id].push_back(ipc_alloc(e->shm, device_id, i, 4, (size_t)num_ids * sizeof(int32_t)));
        e->node_counter[device_id].push_back(ipc_alloc(e->shm, device_id, i, 5, 16 * sizeof(int32_t)));
        e->edge_counter[device_id].push_back(ipc_alloc(e->shm, device_id, i, 6, 16 * sizeof(int32_t)));
        if (i == 0 && e->ext) e->ext->handle_sum[device_id] = handle_checksum(&e->shm->memHandle[device_id][0][0]);
        if (e->node_counter[device_id][i]) HIP_CHECK(hipMemset(e->node_counter[device_id][i], 0, 16 * sizeof(int32_t)));
        if (e->edge_counter[device_id][i]) HIP_CHECK(hipMemset(e->edge_counter[device_id][i], 0, 16 * sizeof(int32_t)));
        // memory lock.  Stale semaphores of a crashed run are removed first (the reference only
        // unlinks in Finalize, CUDA_IPC_Service.cu:319-320, so a crash poisons the next start).
        const std::string ssri = sem_name("r", device_id, i), sswi = sem_name("w", device_id, i);
        sem_unlink(ssri.c_str());
        sem_unlink(sswi.c_str());
        e->semr[device_id][i] = sem_open(ssri.c_str(), O_CREAT | O_RDWR, 0666, 0);
        if (e->semr[device_id][i] == SEM_FAILED) { fprintf(log_file(), "errno = %d\n", errno); e->semr[device_id][i] = nullptr; LEGION_ARG_ERROR("InitializeSamplesBuffer: sem_open(sem_r) failed"); return; }
        e->semw[device_id][i] = sem_open(sswi.c_str(), O_CREAT | O_RDWR, 0666, 0);
        if (e->semw[device_id][i] == SEM_FAILED) { fprintf(log_file(), "errno = %d\n", errno); e->semw[device_id][i] = nullptr; LEGION_ARG_ERROR("InitializeSamplesBuffer: sem_open(sem_w) failed"); return; }
    }
    e->pipeline_depth = pipeline_depth;
}

// InitializeFeaturesBuffer, CUDA_IPC_Service.cu:203-212
void IPCEnv_InitializeFeaturesBuffer(IPCEnv* e, int32_t batch_size, int32_t num_ids, int32_t feature_dim,
                                     int32_t device_id, int32_t pipeline_depth)
{
    (void)batch_size;
    if (!e || device_id < 0 || device_id >= e->device_count) { LEGION_ARG_ERROR("InitializeFeaturesBuffer: bad arguments"); return; }
    if (no_device()) return;
    DeviceGuard guard(device_id);
    const size_t bytes = (size_t)num_ids * feature_dim * sizeof(float);
    const bool vmm = (int64_t)bytes > ipc_max_bytes();
    for (int32_t i = 0; i < pipeline_depth; i++) {
        void* p = nullptr;
        if (vmm) {   // too large for one IPC handle under the PyTorch runtime: chunks + one contiguous mapping on both sides
            VmmRegion* r = vmm_create(device_id, i, bytes);
            if (!r) {
                if (!error_is_fatal()) return;
                fflush(stderr);
                exit(EXIT_FAILURE);
            }
            e->vmm.push_back(r);
            memcpy((void*)&e->shm->memHandle[device_id][i][1], &r->desc, sizeof(r->desc));
            p = r->va;
        } else {
            p = ipc_alloc(e->shm, device_id, i, 1, bytes);
        }
        if (!p) return;   // refused or out of memory: the error is sticky, nothing is registered
        e->float_features[device_id].push_back(p);
    }
    e->ext->feature_rows[device_id] = num_ids;
}
// the row capacity published to the trainers of a device (a server that re-sizes its feature buffers; tests)
void IPCEnv_SetFeatureRows(IPCEnv* e, int32_t device_id, int32_t rows)
{
    if (e && e->ext && device_id >= 0 && device_id < e->device_count) e->ext->feature_rows[device_id] = rows;
}
// nc[word] of the batch about to be posted on (dev, pipe), from the host mirror IPCEnv_MirrorCounters queued (the caller has waited for
// that copy); -1 when the mirror of this batch was not queued
int32_t IPCEnv_MirroredNodeCounter(IPCEnv* e, int32_t dev_id, int32_t current_pipe, int32_t word)
{
    if (!e || !e->ext || dev_id < 0 || dev_id >= e->device_count || word < 0 || word >= 16) return -1;
    const int q = current_pipe % e->pipeline_depth;
    if (!e->mirror_fresh[dev_id][q]) return -1;
    if (!e->shm_pinned) return e->mirror_stage[dev_id][q] ? e->mirror_stage[dev_id][q][word] : -1;
    return e->ext->counters[dev_id][q][word];
}

int32_t IPCEnv_GetRawBatchsize(IPCEnv* e) { return e->raw_batch_size; }

// GetLocalBatchId, CUDA_IPC_Service.cu:219-233
int32_t IPCEnv_GetLocalBatchId(IPCEnv* e, int32_t global_batch_id)
{
    int32_t local_batch_id = -1;
    if (global_batch_id < ((e->train_step + e->valid_step) * e->epoch)) {
        int32_t epoch_batch_id = global_batch_id % (e->train_step + e->valid_step);
        local_batch_id = (epoch_batch_id < e->train_step) ? epoch_batch_id : epoch_batch_id - e->train_step;
    } else {
        local_batch_id = (global_batch_id - ((e->train_step + e->valid_step) * e->epoch)) % e->test_step;
    }
    return local_batch_id;
}

// GetCurrentBatchsize, CUDA_IPC_Service.cu:235-243
int32_t IPCEnv_GetCurrentBatchsize(IPCEnv* e, int32_t dev_id, int32_t current_mode)
{
    if (current_mode == LEGION_TRAINMODE) return e->train_batch_size[dev_id];
    if (current_mode == LEGION_VALIDMODE) return e->valid_batch_size[dev_id];
    return e->test_batch_size[dev_id];
}

// GetCurrentMode, CUDA_IPC_Service.cu:246-259
int32_t IPCEnv_GetCurrentMode(IPCEnv* e, int32_t global_batch_id)
{
    if (global_batch_id < ((e->train_step + e->valid_step) * e->epoch)) {
        int32_t epoch_batch_id = global_batch_id % (e->train_step + e->valid_step);
        return (epoch_batch_id < e->train_step) ? LEGION_TRAINMODE : LEGION_VALIDMODE;
    }
    return LEGION_TESTMODE;
}

#define ENV_GETTER(name, field, type) \
    type* IPCEnv_Get##name(IPCEnv* e, int32_t dev_id, int32_t current_pipe) { \
        if (!e || dev_id < 0 || dev_id >= e->device_count || e->field[dev_id].empty()) return nullptr; \
        return (type*)(e->field[dev_id][current_pipe % e->pipeline_depth]); }
ENV_GETTER(Ids, ids, int32_t)
ENV_GETTER(FloatFeatures, float_features, float)
ENV_GETTER(Labels, labels, int32_t)
ENV_GETTER(AggSrc, agg_src, int32_t)
ENV_GETTER(AggDst, agg_dst, int32_t)
ENV_GETTER(NodeCounter, node_counter, int32_t)
ENV_GETTER(EdgeCounter, edge_counter, int32_t)
#undef ENV_GETTER

// Queue the copy of the pipe's counters into the host mirror on `stream`, behind the kernels that wrote them.  The caller posts the
// pipe only after it has waited for that stream's work (the runner waits for an event recorded behind this call).
void IPCEnv_MirrorCounters(IPCEnv* e, int32_t dev_id, int32_t current_pipe, void* stream)
{
    if (!e || !e->ext || dev_id < 0 || dev_id >= e->device_count || e->node_counter[dev_id].empty() || no_device()) return;
    const int q = current_pipe % e->pipeline_depth;
    int32_t* m = (int32_t*)&e->ext->counters[dev_id][q][0];
    if (!e->shm_pinned) {           // the runtime refused to page-lock the slab: queue the copies into pinned staging words, IPCPost moves them
        if (!e->mirror_stage[dev_id][q] && hipHostMalloc((void**)&e->mirror_stage[dev_id][q], 32 * sizeof(int32_t), hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            e->mirror_stage[dev_id][q] = nullptr;
            return;                 // IPCPost copies synchronously instead
        }
        m = e->mirror_stage[dev_id][q];
    }
    HIP_CHECK(hipMemcpyAsync(m, e->node_counter[dev_id][q], 16 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_CHECK(hipMemcpyAsync(m + 16, e->edge_counter[dev_id][q], 16 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    e->mirror_fresh[dev_id][q] = true;
}
int IPCEnv_SlabPinned(IPCEnv* e) { return e && e->shm_pinned ? 1 : 0; }
// The mirror of a pipe whose counters the HOST decides (a poisoned pipe: nc[*] = -1, ec[*] = 0)
void IPCEnv_SetMirror(IPCEnv* e, int32_t dev_id, int32_t current_pipe, int32_t nc_fill, int32_t ec_fill)
{
    if (!e || !e->ext || dev_id < 0 || dev_id >= e->device_count) return;
    const int q = current_pipe % e->pipeline_depth;
    for (int i = 0; i < 16; i++) { e->ext->counters[dev_id][q][i] = nc_fill; e->ext->counters[dev_id][q][16 + i] = ec_fill; }
    if (e->mirror_stage[dev_id][q]) for (int i = 0; i < 32; i++) e->mirror_stage[dev_id][q][i] = i < 16 ? nc_fill : ec_fill;
    e->mirror_fresh[dev_id][q] = true;
}
void IPCEnv_IPCPost(IPCEnv* e, int32_t dev_id, int32_t current_pipe)
{
    const int q = current_pipe % e->pipeline_depth;
    if (e->ext && e->mirror_fresh[dev_id][q] && !e->shm_pinned && e->mirror_stage[dev_id][q])     // staged by a queued copy the caller has waited for
        for (int i = 0; i < 32; i++) e->ext->counters[dev_id][q][i] = e->mirror_stage[dev_id][q][i];
    if (e->ext && e->ext->mirror_magic == kMirrorMagic && !e->mirror_fresh[dev_id][q] && !e->node_counter[dev_id].empty() && !no_device()) {
        // a producer that did not queue the mirror copy (a reference-style RunOnce on this library): copy now -- the batch is
        // complete when a pipe is posted, so a blocking copy is correct, merely slower than the queued one
        DeviceGuard guard(dev_id);
        int32_t h[32];
        if (hipMemcpy(h, e->node_counter[dev_id][q], 64, hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(h + 16, e->edge_counter[dev_id][q], 64, hipMemcpyDeviceToHost) == hipSuccess)
            for (int i = 0; i < 32; i++) e->ext->counters[dev_id][q][i] = h[i];
        else { (void)hipGetLastError(); e->ext->mirror_magic = 0; }   // clients fall back to the device buffers
    }
    e->mirror_fresh[dev_id][q] = false;
    if (e->semw[dev_id][q]) sem_post(e->semw[dev_id][q]);
}
void IPCEnv_IPCWait(IPCEnv* e, int32_t dev_id, int32_t current_pipe)
{
    sem_t* sem = e->semr[dev_id][current_pipe % e->pipeline_depth];
    if (!sem) return;      // sem_open failed (sticky error): never block on a semaphore that does not exist
    sem_wait_spin(sem);
}
int IPCEnv_HandoffSpinUs(void) { return handoff_spin_us(); }
int IPCEnv_IPCTryWait(IPCEnv* e, int32_t dev_id, int32_t current_pipe, int32_t timeout_ms)
{
    if (timeout_ms <= 0) {     // a pure poll: no clock, no syscall beyond the futex word
        sem_t* sem0 = e->semr[dev_id][current_pipe % e->pipeline_depth];
        return (sem0 && sem_trywait(sem0) == 0) ? 0 : -1;
    }
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_sec += timeout_ms / 1000;
    ts.tv_nsec += (long)(timeout_ms % 1000) * 1000000L;
    if (ts.tv_nsec >= 1000000000L) { ts.tv_sec++; ts.tv_nsec -= 1000000000L; }
    int r;
    sem_t* sem = e->semr[dev_id][current_pipe % e->pipeline_depth];
    if (!sem) return -1;
    while ((r = sem_timedwait(sem, &ts)) != 0 && errno == EINTR) {}
    return r == 0 ? 0 : -1;
}

// Finalize, CUDA_IPC_Service.cu:299-325
void IPCEnv_Finalize(IPCEnv* e)
{
    if (!e) return;
    for (int32_t i = 0; i < e->device_count; i++) {
        if (e->ids[i].empty()) continue;
        LEGION_DEVICE_GUARD(i);
        for (size_t j = 0; j < e->ids[i].size(); j++) {
            if (no_device()) {       // nothing was allocated: only the semaphores exist
                if (e->semw[i][j] && e->semw[i][j] != SEM_FAILED) sem_close(e->semw[i][j]);
                if (e->semr[i][j] && e->semr[i][j] != SEM_FAILED) sem_close(e->semr[i][j]);
                sem_unlink(sem_name("r", i, (int)j).c_str());
                sem_unlink(sem_name("w", i, (int)j).c_str());
                continue;
            }
            (void)hipFree(e->ids[i][j]);
            if (j < e->float_features[i].size()) {
                bool mapped = false;
                for (VmmRegion* r : e->vmm) mapped = mapped || r->va == e->float_features[i][j];
                if (!mapped) (void)hipFree(e->float_features[i][j]);
            }
            (void)hipFree(e->labels[i][j]);
            (void)hipFree(e->agg_src[i][j]);
            (void)hipFree(e->agg_dst[i][j]);
            (void)hipFree(e->node_counter[i][j]);
            (void)hipFree(e->edge_counter[i][j]);
            if (e->semw[i][j] && sem_close(e->semw[i][j]) == -1) log_out() << "close sem " << i << " " << j << " failed\n";
            if (e->semr[i][j]) sem_close(e->semr[i][j]);
            sem_unlink(sem_name("r", i, (int)j).c_str());
            sem_unlink(sem_name("w", i, (int)j).c_str());
        }
        e->ids[i].clear();
    }
    for (VmmRegion* r : e->vmm) { DeviceGuard guard(r->device); vmm_release(r); }
    e->vmm.clear();
    if (e->shm) {
        if (e->shm_pinned && e->ext) { (void)hipHostUnregister((void*)e->ext); e->shm_pinned = false; }
        for (auto& dev : e->mirror_stage) for (auto& p : dev) if (p) { (void)hipHostFree(p); p = nullptr; }
        munmap((void*)e->shm, sizeof(shmStruct));
        close(e->shm_fd);
        shm_unlink(shm_name().c_str());
        e->shm = nullptr;
        if (e->ext) { munmap((void*)e->ext, sizeof(shmExt)); close(e->ext_fd); shm_unlink(ext_name().c_str()); e->ext = nullptr; }
    }
}

int32_t IPCEnv_GetTrainStep(IPCEnv* e) { return e->train_step; }
void IPCEnv_SetHops(IPCEnv* e, int32_t hops) { if (e && e->ext) e->ext->hops = hops; }

} // extern "C"

// ============================ trainer half ===============================================================
struct LegionIPCClient {
    volatile shmStruct* shm = nullptr;
    int shm_fd = -1;
    volatile shmExt* ext = nullptr;    // null: the server has no extension object (a reference server): hops = 2, counters by device copy
    int ext_fd = -1;
    int device = 0;
    void* buf[LEGION_PIPELINE_DEPTH][LEGION_MEMORY_USAGE] = {};
    sem_t* semr[LEGION_PIPELINE_DEPTH] = {};
    sem_t* semw[LEGION_PIPELINE_DEPTH] = {};
    int32_t steps[3] = {0, 0, 0};
    int32_t hops = 2;
    int current_pipe = 0;
    // feature buffers that arrived as chunk descriptors (see VmmDesc): what to unmap / release on close
    struct Mapped { void* va = nullptr; size_t total = 0; std::vector<hipMemGenericAllocationHandle_t> handles; };
    Mapped mapped[LEGION_PIPELINE_DEPTH];
};

// trainer side of a chunked hand-off buffer: fetch the descriptors, import, map back to back
static void* vmm_attach(LegionIPCClient* c, int pipe, const VmmDesc& want)
{
    sockaddr_un addr;
    const socklen_t len = vmm_sock_addr(&addr, c->device, pipe);
    const int s = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    bool ok = s >= 0;
    const char* te = getenv("LEGION_VMM_ATTACH_TIMEOUT_MS");
    const int tries = std::max(1, (te && atoi(te) > 0 ? atoi(te) : 10000) / 100);
    for (int i = 0; ok && connect(s, (sockaddr*)&addr, len) != 0; i++) {
        if (i >= tries - 1) { ok = false; break; }
        usleep(100000);
    }
    if (ok) { timeval tv{10, 0}; (void)setsockopt(s, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv)); }
    VmmDesc d{};
    ok = ok && recv(s, &d, sizeof(d), MSG_WAITALL) == (ssize_t)sizeof(d) && memcmp(&d, &want, sizeof(d)) == 0;
    LegionIPCClient::Mapped& m = c->mapped[pipe];
    int cur = 0;
    (void)hipGetDevice(&cur);
    ok = ok && hipMemAddressReserve(&m.va, d.total, 0, nullptr, 0) == hipSuccess;
    if (ok) m.total = d.total;
    for (uint32_t k = 0; ok && k < d.nchunks; k++) {
        const int fd = recv_fd(s);
        hipMemGenericAllocationHandle_t h;
        ok = fd >= 0 && vmm_import_fd(&h, fd) == hipSuccess;
        if (fd >= 0) close(fd);
        if (!ok) break;
        m.handles.push_back(h);
        const size_t sz = std::min<size_t>(d.chunk, d.total - (size_t)k * d.chunk);
        ok = hipMemMap((char*)m.va + (size_t)k * d.chunk, sz, 0, h, 0) == hipSuccess;
    }
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = cur; acc.flags = hipMemAccessFlagsProtReadWrite;
    ok = ok && hipMemSetAccess(m.va, d.total, &acc, 1) == hipSuccess;
    if (s >= 0) { if (ok) (void)!send(s, "d", 1, MSG_NOSIGNAL); close(s); }
    if (!ok) {
        (void)hipGetLastError();
        // give back whatever was reserved / imported / mapped so far: the caller refuses the whole client
        if (m.va) {
            size_t off = 0;
            for (size_t k = 0; k < m.handles.size(); k++) {
                const size_t sz = std::min<size_t>(d.chunk, d.total - off);
                (void)hipMemUnmap((char*)m.va + off, sz);        // fails harmlessly for the one chunk that was imported but not mapped
                off += sz;
            }
            for (auto h : m.handles) (void)hipMemRelease(h);
            (void)hipMemAddressFree(m.va, m.total);
            (void)hipGetLastError();
        }
        m = LegionIPCClient::Mapped();
        legion_clear_error();   // this is why the client is refused: let it be the error the caller reads
        LEGION_ARG_ERROR("legion_ipc_client_open: attaching the chunked feature buffer failed");
        return nullptr;
    }
    return m.va;
}

extern "C" {

// GPUIPCEnv::Initialize, ipc_cuda_kernel.cu:38-96.  device_id < 0: use the current device
// (the reference reads cudaGetDevice(), set by torch.cuda.set_device(rank) in the trainer).
LegionIPCClient* legion_ipc_client_open(int32_t device_id)
{
    LegionIPCClient* c = new LegionIPCClient();
    int cur = 0;
    if (!no_device()) HIP_CHECK(hipGetDevice(&cur));
    c->device = device_id >= 0 ? device_id : cur;
    // $LEGION_IPC_DEVICE: logical GPU (row of the shm handle table) when it differs from the physical device,
    // e.g. several logical GPUs of a clique exercised on one physical device
    if (device_id < 0 && getenv("LEGION_IPC_DEVICE")) c->device = atoi(getenv("LEGION_IPC_DEVICE"));
    c->shm = (volatile shmStruct*)shm_map(sizeof(shmStruct), &c->shm_fd);
    if (!c->shm) { fprintf(log_file(), "Failed to create shared memory slab\n"); delete c; LEGION_ARG_ERROR("legion_ipc_client_open: shm"); return nullptr; }
    c->ext = (volatile shmExt*)shm_map(sizeof(shmExt), &c->ext_fd, ext_name(), false);     // never created by a client
    for (int i = 0; i < 3; i++) c->steps[i] = c->shm->steps[i];
    if (c->device >= LEGION_MAX_DEVICE) { LEGION_ARG_ERROR("legion_ipc_client_open: device id >= 8"); delete c; return nullptr; }
    // an extension object that does not belong to the slab we attached to (left behind by a killed server, the slab re-created by a server
    // without the extension): ignore it -- 2 hops, counters by device copy, no row bound: the reference's behaviour
    if (c->ext && (c->ext->steps_copy[0] != c->steps[0] || c->ext->steps_copy[1] != c->steps[1] || c->ext->steps_copy[2] != c->steps[2] ||
                   c->ext->handle_sum[c->device] != handle_checksum(&c->shm->memHandle[c->device][0][0]))) {
        munmap((void*)c->ext, sizeof(shmExt));
        close(c->ext_fd);
        c->ext = nullptr; c->ext_fd = -1;
    }
    c->hops = (c->ext && c->ext->hops > 0) ? c->ext->hops : 2;
    // feature buffers that arrive as chunk descriptors first: they need the server's socket, the one step that depends on a second party
    for (int i = 0; i < LEGION_PIPELINE_DEPTH; i++) {
        hipIpcMemHandle_t h;
        memcpy(&h, (const void*)&c->shm->memHandle[c->device][i][1], sizeof(h));
        if (memcmp(&h, kVmmMagic, sizeof(kVmmMagic)) != 0) continue;
        VmmDesc d;
        memcpy(&d, &h, sizeof(d));
        c->buf[i][1] = vmm_attach(c, i, d);
        if (!c->buf[i][1]) {   // no trainer may run on a null feature buffer: close what was opened, post nothing
            legion_ipc_client_close(c);
            return nullptr;
        }
    }
    for (int i = 0; i < LEGION_PIPELINE_DEPTH; i++) {
        for (int w = 0; w < LEGION_MEMORY_USAGE; w++) {
            if (c->buf[i][w]) continue;      // attached above
            hipIpcMemHandle_t h;
            memcpy(&h, (const void*)&c->shm->memHandle[c->device][i][w], sizeof(h));
            static const hipIpcMemHandle_t zero{};
            if (memcmp(&h, &zero, sizeof(h)) == 0) {
                if (no_device()) continue;                       // the device-free test mode registers no buffers at all
                // The server has not registered this buffer yet: the sample buffers appear in Runner_Initialize, the FEATURE buffers only
                // after the pre-sampling epoch (Server.cu:33,273-282).  A trainer that attaches now would run on a null buffer and fault
                // on the GPU in its first kernel; the reference fails in cudaIpcOpenMemHandle here.  Refuse, by name.
                char msg[256];
                snprintf(msg, sizeof(msg), "legion_ipc_client_open: the server has not registered buffer %d of pipe %d of GPU %d yet "
                         "(start the trainer after \"System is ready for serving\")", w, i, c->device);
                legion_ipc_client_close(c);
                legion_clear_error();   // this is why the client is refused: let it be the error the caller reads
                LEGION_ARG_ERROR(msg);
                return nullptr;
            }
            HIP_CHECK(hipIpcOpenMemHandle(&c->buf[i][w], h, hipIpcMemLazyEnablePeerAccess));
        }
    }
    log_out() << "HIP: " << c->device << " IPC shared memory opened\n";
    for (int i = 0; i < LEGION_PIPELINE_DEPTH; i++) {
        c->semr[i] = sem_open(sem_name("r", c->device, i).c_str(), O_CREAT | O_RDWR, 0666, 0);
        if (c->semr[i] == SEM_FAILED) { fprintf(log_file(), "errno = %d\n", errno); LEGION_ARG_ERROR("legion_ipc_client_open: sem_open"); legion_ipc_client_close(c); return nullptr; }
        c->semw[i] = sem_open(sem_name("w", c->device, i).c_str(), O_CREAT | O_RDWR, 0666, 0);
        if (c->semw[i] == SEM_FAILED) { fprintf(log_file(), "errno = %d\n", errno); LEGION_ARG_ERROR("legion_ipc_client_open: sem_open"); legion_ipc_client_close(c); return nullptr; }
        sem_post(c->semr[i]); // both pipes start free (ipc_cuda_kernel.cu:91)
    }
    c->current_pipe = 0;
    return c;
}

void legion_ipc_client_wait(LegionIPCClient* c)
{
    sem_wait_spin(c->semw[c->current_pipe]);
}
// Post(): the pipe's buffers go back to the server, which overwrites them with batch i + 2.  The reference's trainer never
// synchronises its device around synchronize() (legion_graphsage.py:93-116): what bounded its run-ahead was the BLOCKING counter copy of
// the next get_next (ipc_cuda_kernel.cu:195-196, legacy default stream: waits for everything the trainer has queued).  Here get_next reads
// the host mirror and synchronises nothing, so the wait moves to where it is needed: everything this process has queued on its device --
// kernels still reading the batch's ids / rows / COO -- completes BEFORE the semaphore is posted.  One device synchronisation per batch,
// as in the reference, and the pipe is really free when the server sees it.
void legion_ipc_client_post(LegionIPCClient* c)
{
    if (!no_device()) HIP_CHECK(hipDeviceSynchronize());
    legion_ipc_client_post_nosync(c);
}
// ... for a consumer that has already waited for its own work (an event / stream synchronisation of its own, or no device work at all)
void legion_ipc_client_post_nosync(LegionIPCClient* c)
{
    sem_post(c->semr[c->current_pipe]);
    c->current_pipe = (c->current_pipe + 1) % LEGION_PIPELINE_DEPTH;
}
void* legion_ipc_client_buffer(LegionIPCClient* c, int32_t which)
{
    if (!c || which < 0 || which >= LEGION_MEMORY_USAGE) return nullptr;
    return c->buf[c->current_pipe][which];
}
void legion_ipc_client_steps(LegionIPCClient* c, int32_t steps[3])
{
    for (int i = 0; i < 3; i++) steps[i] = c->steps[i];
}
int32_t legion_ipc_client_hops(LegionIPCClient* c) { return c->hops; }
int32_t legion_ipc_client_feature_rows(LegionIPCClient* c) { return (c && c->ext) ? c->ext->feature_rows[c->device] : 0; }
void legion_ipc_client_read_counters(LegionIPCClient* c, int32_t h_node_counter[16], int32_t h_edge_counter[16])
{
    if (c->ext && c->ext->mirror_magic == kMirrorMagic) {
        // the server's host mirror of this pipe: no device copy, no implicit synchronisation with the trainer's own GPU work
        const volatile int32_t* m = &c->ext->counters[c->device][c->current_pipe][0];
        for (int i = 0; i < 16; i++) { h_node_counter[i] = m[i]; h_edge_counter[i] = m[16 + i]; }
        return;
    }
    HIP_CHECK(hipMemcpy(h_node_counter, c->buf[c->current_pipe][5], 16 * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(h_edge_counter, c->buf[c->current_pipe][6], 16 * sizeof(int32_t), hipMemcpyDeviceToHost));
}
void legion_ipc_client_close(LegionIPCClient* c)
{
    if (!c) return;
    for (int i = 0; i < LEGION_PIPELINE_DEPTH; i++) {
        for (int w = 0; w < LEGION_MEMORY_USAGE; w++) {
            if (!c->buf[i][w]) continue;
            if (w == 1 && c->mapped[i].va == c->buf[i][w]) {
                (void)hipMemUnmap(c->mapped[i].va, c->mapped[i].total);
                for (auto h : c->mapped[i].handles) (void)hipMemRelease(h);
                (void)hipMemAddressFree(c->mapped[i].va, c->mapped[i].total);
            } else {
                (void)hipIpcCloseMemHandle(c->buf[i][w]);
            }
        }
        if (c->semw[i] && c->semw[i] != SEM_FAILED && sem_close(c->semw[i]) == -1) log_out() << "close sem " << i << " failed\n";
        if (c->semr[i] && c->semr[i] != SEM_FAILED) sem_close(c->semr[i]);
    }
    if (c->shm) { munmap((void*)c->shm, sizeof(shmStruct)); close(c->shm_fd); }
    if (c->ext) { munmap((void*)c->ext, sizeof(shmExt)); close(c->ext_fd); }
    delete c;
}

} // extern "C"
