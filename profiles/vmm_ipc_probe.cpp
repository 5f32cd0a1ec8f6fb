// vmm_ipc_probe.cpp -- can a hand-off buffer larger than the 2^31-byte HIP-IPC limit of the torch-bundled runtime be shared as a
// list of <= 1 GiB physical chunks (HIP virtual memory management: hipMemCreate + hipMemExportToShareableHandle, POSIX fds passed
// over a unix socket) that the consumer maps into ONE contiguous virtual range?
//   hipcc -O2 --offload-arch=gfx950 profiles/vmm_ipc_probe.cpp -o /tmp/vmm_probe
//   /tmp/vmm_probe export <chunks> <chunk_bytes> <socket path>     (serves one importer, then exits)
//   /tmp/vmm_probe import <chunks> <chunk_bytes> <socket path>     (C++ importer, system runtime)
// profiles/vmm_ipc_probe_torch.py is the importer inside a PyTorch process (bundled runtime).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/socket.h>
#include <sys/un.h>
#include <thread>
#include <unistd.h>
#include <vector>

#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("hip error %s (%d) at line %d\n", hipGetErrorString(r), (int)r, __LINE__); fflush(stdout); _exit(2); } } while (0)

__global__ void fill(unsigned long long* p, size_t n) { for (size_t i = threadIdx.x + (size_t)blockDim.x * blockIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = i * 2654435761ull + 11; }
__global__ void check(const unsigned long long* p, size_t n, unsigned long long* bad) { for (size_t i = threadIdx.x + (size_t)blockDim.x * blockIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) if (p[i] != i * 2654435761ull + 11) atomicAdd(bad, 1ull); }

static int send_fds(int sock, const std::vector<int>& fds)
{
    for (int fd : fds) {
        char byte = 'x';
        iovec io{&byte, 1};
        char ctrl[CMSG_SPACE(sizeof(int))] = {0};
        msghdr msg{}; msg.msg_iov = &io; msg.msg_iovlen = 1; msg.msg_control = ctrl; msg.msg_controllen = sizeof(ctrl);
        cmsghdr* c = CMSG_FIRSTHDR(&msg); c->cmsg_level = SOL_SOCKET; c->cmsg_type = SCM_RIGHTS; c->cmsg_len = CMSG_LEN(sizeof(int));
        memcpy(CMSG_DATA(c), &fd, sizeof(int));
        if (sendmsg(sock, &msg, 0) != 1) return -1;
    }
    return 0;
}
static int recv_fd(int sock)
{
    char byte; iovec io{&byte, 1};
    char ctrl[CMSG_SPACE(sizeof(int))] = {0};
    msghdr msg{}; msg.msg_iov = &io; msg.msg_iovlen = 1; msg.msg_control = ctrl; msg.msg_controllen = sizeof(ctrl);
    if (recvmsg(sock, &msg, 0) != 1) return -1;
    cmsghdr* c = CMSG_FIRSTHDR(&msg);
    int fd = -1; if (c) memcpy(&fd, CMSG_DATA(c), sizeof(int));
    return fd;
}

int main(int argc, char** argv)
{
    if (argc < 5) return 1;
    const bool exporter = !strcmp(argv[1], "export");
    const int chunks = atoi(argv[2]);
    const size_t chunk = strtoull(argv[3], nullptr, 10), total = chunk * chunks;
    const char* path = argv[4];
    CK(hipSetDevice(0));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (chunk % gran) { printf("chunk not a multiple of the granularity %zu\n", gran); return 1; }
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    sockaddr_un addr{}; addr.sun_family = AF_UNIX; strncpy(addr.sun_path, path, sizeof(addr.sun_path) - 1);
    if (exporter) {
        std::vector<int> fds;
        for (int c = 0; c < chunks; c++) {
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, chunk, &prop, 0));
            CK(hipMemMap((char*)va + c * chunk, chunk, 0, h, 0));
            int fd = -1;
            CK(hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0));
            fds.push_back(fd);
        }
        CK(hipMemSetAccess(va, total, &acc, 1));
        fill<<<4096, 256>>>((unsigned long long*)va, total / 8);
        CK(hipDeviceSynchronize());
        unlink(path);
        int ls = socket(AF_UNIX, SOCK_STREAM, 0);
        if (bind(ls, (sockaddr*)&addr, sizeof(addr)) || listen(ls, 1)) { perror("bind/listen"); return 3; }
        int s = accept(ls, nullptr, nullptr);
        if (s < 0 || send_fds(s, fds)) { printf("sending the fds failed\n"); return 3; }
        char done; (void)!read(s, &done, 1);       // importer says when it is finished
        printf("exporter: %d chunks of %zu bytes served (granularity %zu)\n", chunks, chunk, gran);
        return 0;
    }
    int s = socket(AF_UNIX, SOCK_STREAM, 0);
    for (int i = 0; i < 300 && connect(s, (sockaddr*)&addr, sizeof(addr)); i++) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    std::thread([] { std::this_thread::sleep_for(std::chrono::seconds(30)); printf("importer: watchdog after 30 s\n"); fflush(stdout); _exit(3); }).detach();
    auto t0 = std::chrono::steady_clock::now();
    for (int c = 0; c < chunks; c++) {
        int fd = recv_fd(s);
        if (fd < 0) { printf("no fd\n"); return 4; }
        hipMemGenericAllocationHandle_t h;
        CK(hipMemImportFromShareableHandle(&h, (void*)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor));
        CK(hipMemMap((char*)va + c * chunk, chunk, 0, h, 0));
        close(fd);
    }
    CK(hipMemSetAccess(va, total, &acc, 1));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long* bad = nullptr;
    CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    check<<<4096, 256>>>((const unsigned long long*)va, total / 8, bad);
    unsigned long long h_bad = 1;
    CK(hipMemcpy(&h_bad, bad, 8, hipMemcpyDeviceToHost));
    printf("importer: %d x %zu bytes = %.2f GiB mapped contiguously in %.1f ms, %llu wrong words\n", chunks, chunk, total / 1073741824.0, ms, h_bad);
    (void)!write(s, "d", 1);
    return h_bad ? 5 : 0;
}
