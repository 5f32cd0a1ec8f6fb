// ipc_env.cpp -- the server <-> trainer hand-off.
//   server half : CUDAIPCEnv            src/CUDA_IPC_Service.cu:39-360
//   trainer half: GPUIPCEnv             pytorch_extension/ipc_cuda_kernel.cu:36-176
//   shm helpers : helper_multiprocess   src/helper_multiprocess.cpp:5-100
// Contract kept byte for byte: POSIX shm "simpleIPCshm" holding
//   struct { int32 steps[3]; ipcMemHandle[8 dev][2 pipe][7 buf]; }   (12 + 8*2*7*64 = 7180 B;
// hipIpcMemHandle_t is 64 bytes like cudaIpcMemHandle_t), named semaphores sem_r_D_P / sem_w_D_P
// with initial value 0, producer: wait(sem_r) -> fill -> post(sem_w); consumer: wait(sem_w) ->
// use -> post(sem_r).  Extension: one int32 (hop count) appended AFTER the reference struct.
#include "internal.h"

#include <cerrno>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <iostream>
#include <semaphore.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

using namespace legion;

static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle must be 64 bytes (CUDA_IPC_Service.cu:34-37)");

struct shmStruct {
    int32_t steps[3];
    hipIpcMemHandle_t memHandle[LEGION_MAX_DEVICE][LEGION_PIPELINE_DEPTH][LEGION_MEMORY_USAGE];
    int32_t ext_hops; // extension, outside the reference's 7180 bytes
};
static_assert(offsetof(shmStruct, ext_hops) == 12 + 8 * 2 * 7 * 64, "reference shm layout changed");

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
static std::string sem_name(const char* rw, int dev, int pipe)
{
    return "/" + ipc_ns() + "sem_" + rw + "_" + std::to_string(dev) + "_" + std::to_string(pipe);
}

// sharedMemoryCreate (helper_multiprocess.cpp:5-47): shm_open(O_RDWR|O_CREAT) + ftruncate + mmap
static void* shm_map(size_t sz, int* fd_out)
{
    int fd = shm_open(shm_name().c_str(), O_RDWR | O_CREAT, 0777);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) == 0 && (size_t)st.st_size < sz && ftruncate(fd, (off_t)sz) != 0) { close(fd); return nullptr; }
    void* addr = mmap(nullptr, sz, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (addr == MAP_FAILED) { close(fd); return nullptr; }
    *fd_out = fd;
    return addr;
}

struct IPCEnv {
    volatile shmStruct* shm = nullptr;
    int shm_fd = -1;
    int32_t device_count = 0;
    std::vector<std::vector<void*>> ids, float_features, labels, agg_src, agg_dst, node_counter, edge_counter;
    std::vector<std::vector<sem_t*>> semr, semw;
    int32_t raw_batch_size = 0;
    std::vector<int32_t> train_batch_size, valid_batch_size, test_batch_size;
    int32_t train_step = 0, valid_step = 0, test_step = 0, epoch = 0, pipeline_depth = LEGION_PIPELINE_DEPTH;
};

extern "C" {

void legion_ipc_set_namespace(const char* ns)
{
    g_namespace = ns ? ns : "";
    g_ns_init = true;
}

// ============================ server half ==============================================================
// CUDAIPCEnv::CUDAIPCEnv, CUDA_IPC_Service.cu:41-65
IPCEnv* NewIPCEnv(int32_t device_count)
{
    if (device_count < 1 || device_count > LEGION_MAX_DEVICE) { LEGION_ARG_ERROR("NewIPCEnv: device_count must be 1..8"); return nullptr; }
    IPCEnv* e = new IPCEnv();
    std::cout << "start initialize ipc env\n";
    e->shm = (volatile shmStruct*)shm_map(sizeof(shmStruct), &e->shm_fd);
    if (!e->shm) {
        printf("Failed to create shared memory slab\n"); // reference: exit(EXIT_FAILURE) (CUDA_IPC_Service.cu:46-48)
        delete e;
        LEGION_ARG_ERROR("NewIPCEnv: shm_open/mmap failed");
        return nullptr;
    }
    std::cout << "Shared Memory Opened\n";
    // $LEGION_IPC_ATTACH=1: another server process of this job already created the slab
    const char* attach = getenv("LEGION_IPC_ATTACH");
    if (!(attach && attach[0] == '1')) memset((void*)e->shm, 0, sizeof(shmStruct));
    e->device_count = device_count;
    auto rs = [&](std::vector<std::vector<void*>>& v) { v.assign(device_count, {}); };
    rs(e->ids); rs(e->float_features); rs(e->labels); rs(e->agg_src); rs(e->agg_dst); rs(e->node_counter); rs(e->edge_counter);
    e->semr.assign(device_count, {});
    e->semw.assign(device_count, {});
    return e;
}

// Coordinate, CUDA_IPC_Service.cu:66-134
void IPCEnv_Coordinate(IPCEnv* e, const LegionBuildInfo* info)
{
    if (!e || !info) { LEGION_ARG_ERROR("IPCEnv_Coordinate: null argument"); return; }
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
    std::cout << "Train Steps: " << e->train_step << "\n";
    std::cout << "Valid Steps: " << e->valid_step << "\n";
    std::cout << "Test Steps: " << e->test_step << "\n";
    e->shm->steps[0] = e->train_step;
    e->shm->steps[1] = e->valid_step;
    e->shm->steps[2] = e->test_step;
}

int32_t IPCEnv_GetMaxStep(IPCEnv* e) { return ((e->train_step + e->valid_step) * e->epoch) + e->test_step; }

static void* ipc_alloc(volatile shmStruct* shm, int dev, int pipe, int which, size_t bytes)
{
    void* p = nullptr;
    // a trainer would stall forever inside ipc_service.initialize() on a buffer it cannot import
    if (!ipc_size_ok((int64_t)bytes, which == 1 ? "InitializeFeaturesBuffer (rows x F x 4 bytes of one pipe)" : "InitializeSamplesBuffer")) {
        if (!error_is_fatal()) return nullptr;
        fflush(stderr);
        exit(EXIT_FAILURE);
    }
    HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16));
    if (p) HIP_CHECK(hipIpcGetMemHandle((hipIpcMemHandle_t*)&shm->memHandle[dev][pipe][which], p));
    return p;
}

// InitializeSamplesBuffer, CUDA_IPC_Service.cu:140-201
void IPCEnv_InitializeSamplesBuffer(IPCEnv* e, int32_t batch_size, int32_t num_ids, int32_t feature_dim,
                                    int32_t device_id, int32_t pipeline_depth)
{
    (void)feature_dim;
    if (!e || device_id < 0 || device_id >= e->device_count || pipeline_depth < 1 || pipeline_depth > LEGION_PIPELINE_DEPTH) { LEGION_ARG_ERROR("InitializeSamplesBuffer: bad arguments"); return; }
    DeviceGuard guard(device_id);
    e->semr[device_id].assign(pipeline_depth, nullptr);
    e->semw[device_id].assign(pipeline_depth, nullptr);
    for (int32_t i = 0; i < pipeline_depth; i++) {
        e->ids[device_id].push_back(ipc_alloc(e->shm, device_id, i, 0, (size_t)num_ids * sizeof(int32_t)));
        e->labels[device_id].push_back(ipc_alloc(e->shm, device_id, i, 2, (size_t)batch_size * sizeof(int32_t)));
        e->agg_src[device_id].push_back(ipc_alloc(e->shm, device_id, i, 3, (size_t)num_ids * sizeof(int32_t)));
        e->agg_dst[device_id].push_back(ipc_alloc(e->shm, device_id, i, 4, (size_t)num_ids * sizeof(int32_t)));
        e->node_counter[device_id].push_back(ipc_alloc(e->shm, device_id, i, 5, 16 * sizeof(int32_t)));
        e->edge_counter[device_id].push_back(ipc_alloc(e->shm, device_id, i, 6, 16 * sizeof(int32_t)));
        HIP_CHECK(hipMemset(e->node_counter[device_id][i], 0, 16 * sizeof(int32_t)));
        HIP_CHECK(hipMemset(e->edge_counter[device_id][i], 0, 16 * sizeof(int32_t)));
        // memory lock.  Stale semaphores of a crashed run are removed first (the reference only
        // unlinks in Finalize, CUDA_IPC_Service.cu:319-320, so a crash poisons the next start).
        const std::string ssri = sem_name("r", device_id, i), sswi = sem_name("w", device_id, i);
        sem_unlink(ssri.c_str());
        sem_unlink(sswi.c_str());
        e->semr[device_id][i] = sem_open(ssri.c_str(), O_CREAT | O_RDWR, 0666, 0);
        if (e->semr[device_id][i] == SEM_FAILED) { printf("errno = %d\n", errno); return; }
        e->semw[device_id][i] = sem_open(sswi.c_str(), O_CREAT | O_RDWR, 0666, 0);
        if (e->semw[device_id][i] == SEM_FAILED) { printf("errno = %d\n", errno); return; }
    }
    e->pipeline_depth = pipeline_depth;
}

// InitializeFeaturesBuffer, CUDA_IPC_Service.cu:203-212
void IPCEnv_InitializeFeaturesBuffer(IPCEnv* e, int32_t batch_size, int32_t num_ids, int32_t feature_dim,
                                     int32_t device_id, int32_t pipeline_depth)
{
    (void)batch_size;
    if (!e || device_id < 0 || device_id >= e->device_count) { LEGION_ARG_ERROR("InitializeFeaturesBuffer: bad arguments"); return; }
    DeviceGuard guard(device_id);
    for (int32_t i = 0; i < pipeline_depth; i++) {
        void* p = ipc_alloc(e->shm, device_id, i, 1, (size_t)num_ids * feature_dim * sizeof(float));
        if (!p) return;   // refused (size limit) or out of memory: the error is sticky, nothing is registered
        e->float_features[device_id].push_back(p);
    }
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

void IPCEnv_IPCPost(IPCEnv* e, int32_t dev_id, int32_t current_pipe) { sem_post(e->semw[dev_id][current_pipe]); }
void IPCEnv_IPCWait(IPCEnv* e, int32_t dev_id, int32_t current_pipe)
{
    while (sem_wait(e->semr[dev_id][current_pipe]) != 0 && errno == EINTR) {}
}
int IPCEnv_IPCTryWait(IPCEnv* e, int32_t dev_id, int32_t current_pipe, int32_t timeout_ms)
{
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_sec += timeout_ms / 1000;
    ts.tv_nsec += (long)(timeout_ms % 1000) * 1000000L;
    if (ts.tv_nsec >= 1000000000L) { ts.tv_sec++; ts.tv_nsec -= 1000000000L; }
    int r;
    while ((r = sem_timedwait(e->semr[dev_id][current_pipe], &ts)) != 0 && errno == EINTR) {}
    return r == 0 ? 0 : -1;
}

// Finalize, CUDA_IPC_Service.cu:299-325
void IPCEnv_Finalize(IPCEnv* e)
{
    if (!e) return;
    for (int32_t i = 0; i < e->device_count; i++) {
        if (e->ids[i].empty()) continue;
        DeviceGuard guard(i);
        for (size_t j = 0; j < e->ids[i].size(); j++) {
            (void)hipFree(e->ids[i][j]);
            if (j < e->float_features[i].size()) (void)hipFree(e->float_features[i][j]);
            (void)hipFree(e->labels[i][j]);
            (void)hipFree(e->agg_src[i][j]);
            (void)hipFree(e->agg_dst[i][j]);
            (void)hipFree(e->node_counter[i][j]);
            (void)hipFree(e->edge_counter[i][j]);
            if (e->semw[i][j] && sem_close(e->semw[i][j]) == -1) std::cout << "close sem " << i << " " << j << " failed\n";
            if (e->semr[i][j]) sem_close(e->semr[i][j]);
            sem_unlink(sem_name("r", i, (int)j).c_str());
            sem_unlink(sem_name("w", i, (int)j).c_str());
        }
        e->ids[i].clear();
    }
    if (e->shm) {
        munmap((void*)e->shm, sizeof(shmStruct));
        close(e->shm_fd);
        shm_unlink(shm_name().c_str());
        e->shm = nullptr;
    }
}

int32_t IPCEnv_GetTrainStep(IPCEnv* e) { return e->train_step; }
void IPCEnv_SetHops(IPCEnv* e, int32_t hops) { if (e && e->shm) e->shm->ext_hops = hops; }

} // extern "C"

// ============================ trainer half ===============================================================
struct LegionIPCClient {
    volatile shmStruct* shm = nullptr;
    int shm_fd = -1;
    int device = 0;
    void* buf[LEGION_PIPELINE_DEPTH][LEGION_MEMORY_USAGE] = {};
    sem_t* semr[LEGION_PIPELINE_DEPTH] = {};
    sem_t* semw[LEGION_PIPELINE_DEPTH] = {};
    int32_t steps[3] = {0, 0, 0};
    int32_t hops = 2;
    int current_pipe = 0;
};

extern "C" {

// GPUIPCEnv::Initialize, ipc_cuda_kernel.cu:38-96.  device_id < 0: use the current device
// (the reference reads cudaGetDevice(), set by torch.cuda.set_device(rank) in the trainer).
LegionIPCClient* legion_ipc_client_open(int32_t device_id)
{
    LegionIPCClient* c = new LegionIPCClient();
    int cur = 0;
    HIP_CHECK(hipGetDevice(&cur));
    c->device = device_id >= 0 ? device_id : cur;
    // $LEGION_IPC_DEVICE: logical GPU (row of the shm handle table) when it differs from the physical device,
    // e.g. several logical GPUs of a clique exercised on one physical device
    if (device_id < 0 && getenv("LEGION_IPC_DEVICE")) c->device = atoi(getenv("LEGION_IPC_DEVICE"));
    c->shm = (volatile shmStruct*)shm_map(sizeof(shmStruct), &c->shm_fd);
    if (!c->shm) { printf("Failed to create shared memory slab\n"); delete c; LEGION_ARG_ERROR("legion_ipc_client_open: shm"); return nullptr; }
    for (int i = 0; i < 3; i++) c->steps[i] = c->shm->steps[i];
    c->hops = c->shm->ext_hops > 0 ? c->shm->ext_hops : 2;
    if (c->device >= LEGION_MAX_DEVICE) { LEGION_ARG_ERROR("legion_ipc_client_open: device id >= 8"); delete c; return nullptr; }
    for (int i = 0; i < LEGION_PIPELINE_DEPTH; i++) {
        for (int w = 0; w < LEGION_MEMORY_USAGE; w++) {
            hipIpcMemHandle_t h;
            memcpy(&h, (const void*)&c->shm->memHandle[c->device][i][w], sizeof(h));
            HIP_CHECK(hipIpcOpenMemHandle(&c->buf[i][w], h, hipIpcMemLazyEnablePeerAccess));
        }
    }
    std::cout << "HIP: " << c->device << " IPC shared memory opened\n";
    for (int i = 0; i < LEGION_PIPELINE_DEPTH; i++) {
        c->semr[i] = sem_open(sem_name("r", c->device, i).c_str(), O_CREAT | O_RDWR, 0666, 0);
        if (c->semr[i] == SEM_FAILED) { printf("errno = %d\n", errno); LEGION_ARG_ERROR("legion_ipc_client_open: sem_open"); return c; }
        c->semw[i] = sem_open(sem_name("w", c->device, i).c_str(), O_CREAT | O_RDWR, 0666, 0);
        if (c->semw[i] == SEM_FAILED) { printf("errno = %d\n", errno); LEGION_ARG_ERROR("legion_ipc_client_open: sem_open"); return c; }
        sem_post(c->semr[i]); // both pipes start free (ipc_cuda_kernel.cu:91)
    }
    c->current_pipe = 0;
    return c;
}

void legion_ipc_client_wait(LegionIPCClient* c)
{
    while (sem_wait(c->semw[c->current_pipe]) != 0 && errno == EINTR) {}
}
void legion_ipc_client_post(LegionIPCClient* c)
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
void legion_ipc_client_read_counters(LegionIPCClient* c, int32_t h_node_counter[16], int32_t h_edge_counter[16])
{
    HIP_CHECK(hipMemcpy(h_node_counter, c->buf[c->current_pipe][5], 16 * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(h_edge_counter, c->buf[c->current_pipe][6], 16 * sizeof(int32_t), hipMemcpyDeviceToHost));
}
void legion_ipc_client_close(LegionIPCClient* c)
{
    if (!c) return;
    for (int i = 0; i < LEGION_PIPELINE_DEPTH; i++) {
        for (int w = 0; w < LEGION_MEMORY_USAGE; w++)
            if (c->buf[i][w]) (void)hipIpcCloseMemHandle(c->buf[i][w]);
        if (c->semw[i] && c->semw[i] != SEM_FAILED && sem_close(c->semw[i]) == -1) std::cout << "close sem " << i << " failed\n";
        if (c->semr[i] && c->semr[i] != SEM_FAILED) sem_close(c->semr[i]);
    }
    if (c->shm) { munmap((void*)c->shm, sizeof(shmStruct)); close(c->shm_fd); }
    delete c;
}

} // extern "C"
