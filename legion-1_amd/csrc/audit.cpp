// audit.cpp -- the logical-device audit (audit.h has the what and why).  This file calls HIP directly: it never includes audit_hooks.h.
#include "internal.h"

#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>

namespace legion {

static thread_local int t_logical = -1;
int current_logical_device() { return t_logical; }
void set_current_logical_device(int logical) { t_logical = logical; }

namespace audit {

bool g_on = [] { const char* e = getenv("LEGION_DEVICE_AUDIT"); return e && e[0] == '1'; }();

namespace {

struct Site { const char* file; int line; };
struct Alloc {
    uintptr_t base; size_t size;
    int dev;                 // logical GPU it was created under (-1: none was selected)
    uint32_t shared;         // further logical GPUs it serves (they share its physical device)
    char kind;               // 'D' device, 'H' pinned host, 'M' managed, 'I' IPC import, 'V' mapped VMM range
    Site site;
};
struct Tag { int dev; Site site; };

std::mutex mu;
std::map<uintptr_t, Alloc> allocs;                         // by base address
std::unordered_map<void*, Tag> streams, events, execs;
bool peer[64][64];
int canon[64];                                              // alias classes of logical ids (legion_audit_alias); canon[i] == i unless aliased
bool canon_init = [] { for (int i = 0; i < 64; i++) canon[i] = i; return true; }();
inline int cn(int d) { return (d >= 0 && d < 64) ? canon[d] : d; }
bool has_peer(int from, int to)
{
    for (int i = 0; i < 64; i++)
        if (canon[i] == cn(from))
            for (int j = 0; j < 64; j++)
                if (canon[j] == cn(to) && peer[i][j]) return true;
    return false;
}
std::atomic<int64_t> n_checks{0}, n_violations{0}, n_unattributed{0}, n_peer_launches{0};
std::vector<std::string> messages;                          // first kMaxMessages violations
constexpr size_t kMaxMessages = 256;

const char* base_name(const char* path)
{
    const char* s = strrchr(path, '/');
    return s ? s + 1 : path;
}

void violation(const char* file, int line, const std::string& text)
{
    n_violations++;
    const std::string msg = "device audit: " + text;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (messages.size() < kMaxMessages) messages.push_back(std::string(base_name(file)) + ":" + std::to_string(line) + ": " + msg);
    }
    report_error(file, line, msg.c_str(), false);
}

std::string where(const Site& s) { return std::string(base_name(s.file)) + ":" + std::to_string(s.line); }
std::string gpu(int d) { return d < 0 ? std::string("no logical GPU") : "logical GPU " + std::to_string(d); }

// callers hold `mu`
const Alloc* find(const void* p)
{
    if (!p) return nullptr;
    const uintptr_t a = (uintptr_t)p;
    auto it = allocs.upper_bound(a);
    if (it == allocs.begin()) return nullptr;
    --it;
    return a < it->second.base + (it->second.size ? it->second.size : 1) ? &it->second : nullptr;
}

enum Access { kLocal, kHost, kPeer, kImport, kUnknown, kBad };
Access classify(const Alloc* a, int cur)
{
    if (!a) return kUnknown;
    if (a->kind == 'H' || a->kind == 'M') return kHost;
    if (a->dev < 0 || cur < 0 || cur >= 64) return kUnknown;
    if (cn(a->dev) == cn(cur) || ((a->shared >> cur) & 1u)) return kLocal;
    if (a->kind == 'I') return kImport;                     // opened by this process (hipIpcMemLazyEnablePeerAccess)
    if (a->dev < 64 && has_peer(cur, a->dev)) return kPeer;
    return kBad;
}

// HIP's own current device must be the physical device behind the thread's logical one
void check_physical(int cur, const char* what, const char* file, int line)
{
    if (cur < 0) return;
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return; }
    const int want = physical_device(cur);
    if (d != want)
        violation(file, line, std::string(what) + ": the thread selected " + gpu(cur) + " (physical device " + std::to_string(want) +
                                  ") but HIP's current device is " + std::to_string(d) + " (a hipSetDevice outside SetGPUDevice / DeviceGuard?)");
}

// tag of a stream / event / graph (null stream: the current device's); false: not one of ours
bool tag_of(const std::unordered_map<void*, Tag>& m, void* h, int cur, Tag* out)
{
    if (!h) { *out = Tag{cur, {"<null stream>", 0}}; return true; }
    std::lock_guard<std::mutex> lk(mu);
    auto it = m.find(h);
    if (it == m.end()) return false;
    *out = it->second;
    return true;
}

void check_stream(hipStream_t s, const char* what, const char* file, int line)
{
    const int cur = t_logical;
    n_checks++;
    check_physical(cur, what, file, line);
    Tag t;
    if (!tag_of(streams, (void*)s, cur, &t) || t.dev < 0 || cur < 0) { n_unattributed++; return; }
    if (cn(t.dev) != cn(cur))
        violation(file, line, std::string(what) + ": stream of " + gpu(t.dev) + " (created " + where(t.site) + ") used under " + gpu(cur));
}

void add_alloc(void* p, size_t n, char kind, const char* f, int l)
{
    const int cur = t_logical;
    n_checks++;
    if (cur < 0) n_unattributed++;
    else check_physical(cur, "allocation", f, l);
    std::lock_guard<std::mutex> lk(mu);
    allocs[(uintptr_t)p] = Alloc{(uintptr_t)p, n, cur, 0u, kind, {f, l}};
}
void drop_alloc(void* p)
{
    std::lock_guard<std::mutex> lk(mu);
    allocs.erase((uintptr_t)p);
}

void add_tag(std::unordered_map<void*, Tag>& m, void* h, const char* what, const char* f, int l)
{
    const int cur = t_logical;
    n_checks++;
    if (cur < 0) n_unattributed++;
    else check_physical(cur, what, f, l);
    std::lock_guard<std::mutex> lk(mu);
    m[h] = Tag{cur, {f, l}};
}
void drop_tag(std::unordered_map<void*, Tag>& m, void* h)
{
    std::lock_guard<std::mutex> lk(mu);
    m.erase(h);
}

// one end of a copy / memset: legal unless it is another logical GPU's memory without recorded peer access
void check_copy_end(const void* p, const char* role, const char* what, const char* file, int line)
{
    const int cur = t_logical;
    Access acc;
    Alloc a{};
    {
        std::lock_guard<std::mutex> lk(mu);
        const Alloc* f = find(p);
        acc = classify(f, cur);
        if (f) a = *f;
    }
    if (acc == kBad)
        violation(file, line, std::string(what) + ": " + role + " is memory of " + gpu(a.dev) + " (allocated " + where(a.site) + "), touched under " +
                                  gpu(cur) + " without recorded peer access");
}

} // namespace

// ------------------------------------------------------------------------------------------------------------------------
void launch(hipStream_t s, const char* kernel, const char* file, int line, std::initializer_list<Arg> args)
{
    const int cur = t_logical;
    check_stream(s, kernel, file, line);
    bool unknown = false, peer_arg = false;
    for (const Arg& g : args) {
        if (!g.p) continue;
        Access acc;
        Alloc a{};
        {
            std::lock_guard<std::mutex> lk(mu);
            const Alloc* f = find(g.p);
            acc = classify(f, cur);
            if (f) a = *f;
        }
        if (acc == kUnknown) { unknown = true; continue; }
        if (acc == kPeer && g.mode == 'r') { peer_arg = true; continue; }
        if (acc == kBad || acc == kPeer)
            violation(file, line, std::string(kernel) + ": argument " + g.name + " is memory of " + gpu(a.dev) + " (allocated " + where(a.site) + "), " +
                                      (g.mode == 'w' ? "WRITTEN" : g.mode == 'l' ? "used as local scratch" : "read without recorded peer access") + " from " + gpu(cur));
    }
    if (unknown) n_unattributed++;
    if (peer_arg) n_peer_launches++;
}

void table(int viewer, const void* const* ptrs, size_t n, const char* what, const char* file, int line)
{
    n_checks++;
    for (size_t i = 0; i < n; i++) {
        if (!ptrs[i]) continue;
        Access acc;
        Alloc a{};
        {
            std::lock_guard<std::mutex> lk(mu);
            const Alloc* f = find(ptrs[i]);
            acc = classify(f, viewer);
            if (f) a = *f;
        }
        if (acc == kUnknown) n_unattributed++;
        else if (acc == kBad)
            violation(file, line, std::string(what) + ": entry " + std::to_string(i) + " of the table of " + gpu(viewer) + " points into memory of " + gpu(a.dev) +
                                      " (allocated " + where(a.site) + ") without recorded peer access");
    }
}

void expect_owner(const void* p, int logical, const char* what, const char* file, int line)
{
    if (!p) return;
    n_checks++;
    Alloc a{};
    bool known;
    {
        std::lock_guard<std::mutex> lk(mu);
        const Alloc* f = find(p);
        known = f != nullptr;
        if (f) a = *f;
    }
    if (!known || a.dev < 0) { n_unattributed++; return; }
    if (a.kind == 'H' || a.kind == 'M') return;
    if (cn(a.dev) != cn(logical) && !((a.shared >> logical) & 1u))
        violation(file, line, std::string(what) + ": expected memory of " + gpu(logical) + ", got memory of " + gpu(a.dev) + " (allocated " + where(a.site) + ")");
}

void expect_stream(hipStream_t s, int logical, const char* what, const char* file, int line)
{
    if (!s) return;
    n_checks++;
    Tag t;
    if (!tag_of(streams, (void*)s, -1, &t) && !tag_of(events, (void*)s, -1, &t)) { n_unattributed++; return; }
    if (t.dev >= 0 && cn(t.dev) != cn(logical))
        violation(file, line, std::string(what) + ": expected a stream / event of " + gpu(logical) + ", got one of " + gpu(t.dev) + " (created " + where(t.site) + ")");
}

void share(const void* p, int logical)
{
    if (!p || logical < 0 || logical >= 32) return;
    std::lock_guard<std::mutex> lk(mu);
    auto it = allocs.find((uintptr_t)p);
    if (it != allocs.end()) it->second.shared |= 1u << logical;
}

void record_peer(int a, int b)
{
    if (a < 0 || b < 0 || a >= 64 || b >= 64) return;
    std::lock_guard<std::mutex> lk(mu);
    peer[a][b] = true;
}

void region(void* va, size_t bytes, int logical, const char* file, int line)
{
    n_checks++;
    std::lock_guard<std::mutex> lk(mu);
    allocs[(uintptr_t)va] = Alloc{(uintptr_t)va, bytes, logical, 0u, 'V', {file, line}};
}
void region_gone(void* va) { drop_alloc(va); }

// ---- HIP calls ---------------------------------------------------------------------------------------------------------
hipError_t Malloc(void** p, size_t n, const char* f, int l)
{
    const hipError_t e = hipMalloc(p, n);
    if (g_on && e == hipSuccess && *p) add_alloc(*p, n, 'D', f, l);
    return e;
}
hipError_t MallocManaged(void** p, size_t n, const char* f, int l)
{
    const hipError_t e = hipMallocManaged(p, n);
    if (g_on && e == hipSuccess && *p) add_alloc(*p, n, 'M', f, l);
    return e;
}
hipError_t Free(void* p, const char*, int)
{
    if (g_on && p) drop_alloc(p);
    return hipFree(p);
}
hipError_t HostMalloc(void** p, size_t n, unsigned flags, const char* f, int l)
{
    const hipError_t e = hipHostMalloc(p, n, flags);
    if (g_on && e == hipSuccess && *p) {
        add_alloc(*p, n, 'H', f, l);
        void* d = nullptr;      // the device view of a mapped allocation may be another address
        if ((flags & hipHostMallocMapped) && hipHostGetDevicePointer(&d, *p, 0) == hipSuccess && d && d != *p) add_alloc(d, n, 'H', f, l);
        (void)hipGetLastError();
    }
    return e;
}
hipError_t HostFree(void* p, const char*, int)
{
    if (g_on && p) {
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, p, 0) == hipSuccess && d && d != p) drop_alloc(d);
        (void)hipGetLastError();
        drop_alloc(p);
    }
    return hipHostFree(p);
}
hipError_t IpcOpen(void** p, hipIpcMemHandle_t h, unsigned flags, const char* f, int l)
{
    const hipError_t e = hipIpcOpenMemHandle(p, h, flags);
    if (g_on && e == hipSuccess && *p) {
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)*p) != hipSuccess) { (void)hipGetLastError(); size = 1; }
        add_alloc(*p, size ? size : 1, 'I', f, l);
    }
    return e;
}
hipError_t IpcClose(void* p, const char*, int)
{
    if (g_on && p) drop_alloc(p);
    return hipIpcCloseMemHandle(p);
}
hipError_t StreamCreateWithFlags(hipStream_t* s, unsigned flags, const char* f, int l)
{
    const hipError_t e = hipStreamCreateWithFlags(s, flags);
    if (g_on && e == hipSuccess) add_tag(streams, (void*)*s, "stream", f, l);
    return e;
}
hipError_t StreamCreateWithPriority(hipStream_t* s, unsigned flags, int prio, const char* f, int l)
{
    const hipError_t e = hipStreamCreateWithPriority(s, flags, prio);
    if (g_on && e == hipSuccess) add_tag(streams, (void*)*s, "stream", f, l);
    return e;
}
hipError_t StreamCreateWithCUMask(hipStream_t* s, uint32_t words, const uint32_t* mask, const char* f, int l)
{
    const hipError_t e = hipExtStreamCreateWithCUMask(s, words, mask);
    if (g_on && e == hipSuccess) add_tag(streams, (void*)*s, "stream", f, l);
    return e;
}
hipError_t StreamDestroy(hipStream_t s, const char*, int)
{
    if (g_on && s) drop_tag(streams, (void*)s);
    return hipStreamDestroy(s);
}
hipError_t EventCreate(hipEvent_t* ev, const char* f, int l)
{
    const hipError_t e = hipEventCreate(ev);
    if (g_on && e == hipSuccess) add_tag(events, (void*)*ev, "event", f, l);
    return e;
}
hipError_t EventCreateWithFlags(hipEvent_t* ev, unsigned flags, const char* f, int l)
{
    const hipError_t e = hipEventCreateWithFlags(ev, flags);
    if (g_on && e == hipSuccess) add_tag(events, (void*)*ev, "event", f, l);
    return e;
}
hipError_t EventDestroy(hipEvent_t ev, const char*, int)
{
    if (g_on && ev) drop_tag(events, (void*)ev);
    return hipEventDestroy(ev);
}
hipError_t EventRecord(hipEvent_t ev, hipStream_t s, const char* f, int l)
{
    if (g_on) {
        check_stream(s, "hipEventRecord", f, l);
        Tag te, ts;
        const int cur = t_logical;
        if (tag_of(events, (void*)ev, cur, &te) && tag_of(streams, (void*)s, cur, &ts) && te.dev >= 0 && ts.dev >= 0 && cn(te.dev) != cn(ts.dev))
            violation(f, l, "hipEventRecord: event of " + gpu(te.dev) + " (created " + where(te.site) + ") recorded on a stream of " + gpu(ts.dev) +
                                " (HIP refuses that on distinct devices)");
    }
    return hipEventRecord(ev, s);
}
hipError_t StreamWaitEvent(hipStream_t s, hipEvent_t ev, unsigned flags, const char* f, int l)
{
    if (g_on) check_stream(s, "hipStreamWaitEvent", f, l);    // the event may belong to any device: cross-device waits are legal
    return hipStreamWaitEvent(s, ev, flags);
}
hipError_t Memcpy(void* dst, const void* src, size_t n, hipMemcpyKind kind, const char* f, int l)
{
    if (g_on) {
        n_checks++;
        check_physical(t_logical, "hipMemcpy", f, l);
        check_copy_end(dst, "the destination", "hipMemcpy", f, l);
        check_copy_end(src, "the source", "hipMemcpy", f, l);
    }
    return hipMemcpy(dst, src, n, kind);
}
hipError_t MemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s, const char* f, int l)
{
    if (g_on) {
        check_stream(s, "hipMemcpyAsync", f, l);
        check_copy_end(dst, "the destination", "hipMemcpyAsync", f, l);
        check_copy_end(src, "the source", "hipMemcpyAsync", f, l);
    }
    return hipMemcpyAsync(dst, src, n, kind, s);
}
hipError_t Memcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t w, size_t h, hipMemcpyKind kind, const char* f, int l)
{
    if (g_on) {
        n_checks++;
        check_physical(t_logical, "hipMemcpy2D", f, l);
        check_copy_end(dst, "the destination", "hipMemcpy2D", f, l);
        check_copy_end(src, "the source", "hipMemcpy2D", f, l);
    }
    return hipMemcpy2D(dst, dpitch, src, spitch, w, h, kind);
}
hipError_t MemcpyPeerAsync(void* dst, int ddev, const void* src, int sdev, size_t n, hipStream_t s, const char* f, int l)
{
    if (g_on) {
        check_stream(s, "hipMemcpyPeerAsync", f, l);
        const struct { const void* p; int phys; const char* role; } ends[2] = {{dst, ddev, "the destination"}, {src, sdev, "the source"}};
        int owner[2] = {-1, -1};
        for (int i = 0; i < 2; i++) {
            Alloc a{};
            bool known;
            {
                std::lock_guard<std::mutex> lk(mu);
                const Alloc* fa = find(ends[i].p);
                known = fa != nullptr;
                if (fa) a = *fa;
            }
            if (!known || a.dev < 0) { n_unattributed++; continue; }
            owner[i] = a.dev;
            if (a.kind != 'H' && a.kind != 'M' && physical_device(a.dev) != ends[i].phys)
                violation(f, l, std::string("hipMemcpyPeerAsync: ") + ends[i].role + " is memory of " + gpu(a.dev) + " (physical device " +
                                    std::to_string(physical_device(a.dev)) + ", allocated " + where(a.site) + ") but the call names device " + std::to_string(ends[i].phys));
        }
        const int cur = t_logical;
        if (cur >= 0 && owner[0] >= 0 && owner[1] >= 0 && cn(cur) != cn(owner[0]) && cn(cur) != cn(owner[1]))
            violation(f, l, "hipMemcpyPeerAsync between " + gpu(owner[1]) + " and " + gpu(owner[0]) + " queued under " + gpu(cur));
    }
    return hipMemcpyPeerAsync(dst, ddev, src, sdev, n, s);
}
hipError_t Memset(void* dst, int v, size_t n, const char* f, int l)
{
    if (g_on) {
        n_checks++;
        check_physical(t_logical, "hipMemset", f, l);
        check_copy_end(dst, "the destination", "hipMemset", f, l);
    }
    return hipMemset(dst, v, n);
}
hipError_t MemsetAsync(void* dst, int v, size_t n, hipStream_t s, const char* f, int l)
{
    if (g_on) {
        check_stream(s, "hipMemsetAsync", f, l);
        check_copy_end(dst, "the destination", "hipMemsetAsync", f, l);
    }
    return hipMemsetAsync(dst, v, n, s);
}
hipError_t GraphInstantiate(hipGraphExec_t* exec, hipGraph_t g, hipGraphNode_t* en, char* log, size_t n, const char* f, int l)
{
    const hipError_t e = hipGraphInstantiate(exec, g, en, log, n);
    if (g_on && e == hipSuccess) add_tag(execs, (void*)*exec, "graph", f, l);
    return e;
}
hipError_t GraphLaunch(hipGraphExec_t exec, hipStream_t s, const char* f, int l)
{
    if (g_on) {
        check_stream(s, "hipGraphLaunch", f, l);
        Tag t;
        const int cur = t_logical;
        if (tag_of(execs, (void*)exec, cur, &t) && t.dev >= 0 && cur >= 0 && cn(t.dev) != cn(cur))
            violation(f, l, "hipGraphLaunch: graph recorded under " + gpu(t.dev) + " (instantiated " + where(t.site) + ") launched under " + gpu(cur));
    }
    return hipGraphLaunch(exec, s);
}
hipError_t GraphExecDestroy(hipGraphExec_t exec, const char*, int)
{
    if (g_on && exec) drop_tag(execs, (void*)exec);
    return hipGraphExecDestroy(exec);
}
hipError_t StreamBeginCapture(hipStream_t s, hipStreamCaptureMode mode, const char* f, int l)
{
    if (g_on) check_stream(s, "hipStreamBeginCapture", f, l);
    return hipStreamBeginCapture(s, mode);
}

} // namespace audit
} // namespace legion

using namespace legion;

extern "C" {

int legion_audit_enabled(void) { return audit::g_on ? 1 : 0; }
void legion_audit_counts(int64_t out[4])
{
    if (!out) return;
    out[0] = audit::n_checks.load(); out[1] = audit::n_violations.load();
    out[2] = audit::n_unattributed.load(); out[3] = audit::n_peer_launches.load();
}
int32_t legion_audit_message_count(void)
{
    std::lock_guard<std::mutex> lk(audit::mu);
    return (int32_t)audit::messages.size();
}
const char* legion_audit_message(int32_t i)
{
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(audit::mu);
    if (i < 0 || (size_t)i >= audit::messages.size()) return "";
    copy = audit::messages[(size_t)i];
    return copy.c_str();
}
// One process per GPU: the process's one device is logical GPU 0 of its replicated engine and logical GPU <rank> of a clique engine.
// Declares that a and b name the SAME device in this process (both must map to one physical device); their tags then compare equal.
int legion_audit_alias(int32_t a, int32_t b)
{
    if (a < 0 || b < 0 || a >= 64 || b >= 64 || physical_device(a) != physical_device(b)) {
        LEGION_ARG_ERROR("legion_audit_alias: two logical ids in [0, 64) that map to the same physical device");
        return -1;
    }
    std::lock_guard<std::mutex> lk(audit::mu);
    const int from = audit::canon[b], to = audit::canon[a];
    for (int i = 0; i < 64; i++) if (audit::canon[i] == from) audit::canon[i] = to;
    return 0;
}
void legion_audit_reset(void)
{
    audit::n_checks = 0; audit::n_violations = 0; audit::n_unattributed = 0; audit::n_peer_launches = 0;
    std::lock_guard<std::mutex> lk(audit::mu);
    audit::messages.clear();
}
// one line for a log; returns the number of violations
int64_t legion_audit_report(void)
{
    if (!audit::g_on) return 0;
    int64_t c[4];
    legion_audit_counts(c);
    fprintf(log_file(), "Device audit: %lld checks, %lld violations, %lld unattributed, %lld launches with peer arguments\n", (long long)c[0], (long long)c[1],
            (long long)c[2], (long long)c[3]);
    const int32_t n = legion_audit_message_count();
    for (int32_t i = 0; i < n && i < 20; i++) fprintf(log_file(), "  %s\n", legion_audit_message(i));
    fflush(log_file());
    return c[1];
}

} // extern "C"
