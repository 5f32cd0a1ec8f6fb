// storage.cpp -- host mirrors of GPUMemoryGraphStorage (GPU_Memory_Graph_Storage.cu:37-210),
// GPUMemoryNodeStorage (GPU_Memory_Node_Storage.cu:3-207) and GPUMemoryPool (GPUMemoryPool.cuh:7-208).
#include "internal.h"

#include <algorithm>
#include <cstring>
#include <set>

#include "audit_hooks.h"

using namespace legion;

// all-pairs peer access between the physical devices behind the logical GPUs
// (GPUGraphStore::EnableP2PAccess, GPUGraphStore.cu:145-168)
static void enable_p2p(int32_t partition_count)
{
    std::set<int> phys;
    for (int i = 0; i < partition_count; i++) phys.insert(physical_device(i));
    // the audit's view: on a node every logical GPU is a device of its own and this loop enables every pair
    if (audit::on()) for (int a = 0; a < partition_count; a++) for (int b = 0; b < partition_count; b++) if (a != b) audit::record_peer(a, b);
    if (phys.size() < 2) return;
    int cur = 0;
    HIP_CHECK(hipGetDevice(&cur));
    for (int a : phys) {
        HIP_CHECK(hipSetDevice(a));
        for (int b : phys) {
            if (a == b) continue;
            int ok = 0;
            HIP_CHECK(hipDeviceCanAccessPeer(&ok, a, b));
            if (ok) {
                hipError_t e = hipDeviceEnablePeerAccess(b, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_CHECK(e);
                (void)hipGetLastError();
            }
        }
    }
    HIP_CHECK(hipSetDevice(cur));
}

template <typename T>
static T* adopt_table(const T* src, int64_t count, int32_t location, bool* owns)
{
    *owns = false;
    if (location == LEGION_LOC_HOST_PAGEABLE) {
        T* p = (T*)host_alloc_space64(count * (int64_t)sizeof(T));
        if (p) memcpy(p, src, (size_t)count * sizeof(T));
        *owns = true;
        return p;
    }
    return const_cast<T*>(src);
}

// one HBM replica per distinct physical device among the local logical GPUs
template <typename T>
static void replicate_table(const T* src, int64_t count, int32_t P, std::vector<T*>& replica)
{
    std::vector<std::pair<int, T*>> per_phys;
    for (int p = 0; p < P; p++) {
        if (is_remote_device(p) || replica[p]) continue;
        const int phys = physical_device(p);
        T* have = nullptr;
        for (auto& e : per_phys) if (e.first == phys) have = e.second;
        if (!have) {
            DeviceGuard guard(p);
            HIP_CHECK(hipMalloc(&have, (size_t)count * sizeof(T)));
            if (have) HIP_CHECK(hipMemcpy(have, src, (size_t)count * sizeof(T), hipMemcpyDefault));
            per_phys.emplace_back(phys, have);
        }
        LEGION_AUDIT_SHARE(have, p);     // logical GPUs of one physical device share its replica
        replica[p] = have;
    }
}
template <typename T>
static void free_replicas(std::vector<T*>& replica)
{
    for (size_t i = 0; i < replica.size(); i++) {
        if (!replica[i]) continue;
        T* p = replica[i];
        for (size_t j = i; j < replica.size(); j++) if (replica[j] == p) replica[j] = nullptr;
        (void)hipFree(p);
    }
}

extern "C" {

// ================================= graph storage ====================================================
GPUGraphStorage* NewGPUMemoryGraphStorage(void) { return new GPUGraphStorage(); }

void GPUGraphStorage_Build(GPUGraphStorage* g, const LegionBuildInfo* info)
{
    if (!g || !info) { LEGION_ARG_ERROR("GPUGraphStorage_Build: null argument"); return; }
    if (info->partition_count < 1 || info->partition_count > kMaxParts) { LEGION_ARG_ERROR("GPUGraphStorage_Build: partition_count must be 1..8"); return; }
    const int P = info->partition_count;
    g->partition_count = P;
    g->node_num = info->total_num_nodes;
    g->edge_num = info->total_edge_num;
    g->cache_edge_num = info->cache_edge_num;
    g->csr_location = info->csr_location;
    enable_p2p(P);
    bool o1 = false, o2 = false;
    g->csr_node_index_cpu = adopt_table<int64_t>(info->csr_node_index, (int64_t)info->total_num_nodes + 1, info->csr_location, &o1);
    g->csr_dst_node_ids_cpu = adopt_table<int32_t>(info->csr_dst_node_ids, info->total_edge_num, info->csr_location, &o2);
    g->owns_csr = o1 || o2;
    if (o1) g->csr_location = LEGION_LOC_HOST_PINNED;
    g->frag.assign(P, GPUGraphStorage::Fragment());
    g->replica_indptr.assign(P, nullptr);
    g->replica_indices.assign(P, nullptr);
    g->view.assign(P, std::vector<bool>(P, false));
    g->d_frag_tab.assign(P, nullptr);
    // chunk geometry of the fragments: powers of two that fit $LEGION_SHARD_CHUNK_BYTES (default 1 GiB)
    const char* env_chunk = getenv("LEGION_SHARD_CHUNK_BYTES");
    const int64_t chunk_bytes = env_chunk ? atoll(env_chunk) : (1ll << 30);
    g->row_shift = 4; g->edge_shift = 4;
    while (g->row_shift < 30 && (2ll << g->row_shift) * (int64_t)sizeof(int64_t) <= chunk_bytes) g->row_shift++;
    while (g->edge_shift < 30 && (2ll << g->edge_shift) * (int64_t)sizeof(int32_t) <= chunk_bytes) g->edge_shift++;
}

static int frag_ip_chunks(const GPUGraphStorage* g, int32_t rows) { return rows > 0 ? (int)((((int64_t)rows - 1) >> g->row_shift) + 1) : 1; }
static int frag_ix_chunks(const GPUGraphStorage* g, int64_t edges) { return edges > 0 ? (int)(((edges - 1) >> g->edge_shift) + 1) : 1; }

static void free_fragment(GPUGraphStorage::Fragment& f)
{
    for (int64_t* p : f.ip) if (p) { if (f.imported) (void)hipIpcCloseMemHandle(p); else (void)hipFree(p); }
    for (int32_t* p : f.ix) if (p) { if (f.imported) (void)hipIpcCloseMemHandle(p); else (void)hipFree(p); }
    f = GPUGraphStorage::Fragment();
}

static bool fragment_complete(const GPUGraphStorage* g, int dev)
{
    const auto& f = g->frag[dev];
    if (f.rows <= 0 || (int)f.ip.size() != frag_ip_chunks(g, f.rows) || (int)f.ix.size() != frag_ix_chunks(g, f.edges)) return false;
    for (auto* p : f.ip) if (!p) return false;
    for (auto* p : f.ix) if (!p) return false;
    return true;
}

// (re)write the device-side chunk-pointer tables of every local viewer
static void publish_fragment_tables(GPUGraphStorage* g)
{
    const int P = g->partition_count;
    int ip_nch = 1, ix_nch = 1;
    for (int p = 0; p < P; p++) { ip_nch = std::max(ip_nch, (int)g->frag[p].ip.size()); ix_nch = std::max(ix_nch, (int)g->frag[p].ix.size()); }
    const bool regrow = ip_nch != g->ip_nch || ix_nch != g->ix_nch;
    g->ip_nch = ip_nch; g->ix_nch = ix_nch;
    std::vector<void*> h((size_t)P * (ip_nch + ix_nch) + 1); // +1: a zero-degree row at offset == edges names chunk ix_nch
    for (int dev = 0; dev < P; dev++) {
        if (is_remote_device(dev)) continue;
        bool any = false;
        std::fill(h.begin(), h.end(), nullptr);
        for (int p = 0; p < P; p++) {
            if (!g->view[dev][p]) continue;
            for (size_t q = 0; q < g->frag[p].ip.size(); q++) { h[(size_t)p * ip_nch + q] = g->frag[p].ip[q]; any = true; }
            for (size_t q = 0; q < g->frag[p].ix.size(); q++) h[(size_t)P * ip_nch + (size_t)p * ix_nch + q] = g->frag[p].ix[q];
        }
        DeviceGuard guard(dev);
        LEGION_AUDIT_TABLE(dev, h.data(), h.size(), "fragment chunk table");
        if (g->d_frag_tab[dev] && (regrow || !any)) { HIP_CHECK(hipDeviceSynchronize()); (void)hipFree(g->d_frag_tab[dev]); g->d_frag_tab[dev] = nullptr; }
        if (!any) continue;
        if (!g->d_frag_tab[dev]) HIP_CHECK(hipMalloc(&g->d_frag_tab[dev], h.size() * sizeof(void*)));
        HIP_CHECK(hipMemcpy(g->d_frag_tab[dev], h.data(), h.size() * sizeof(void*), hipMemcpyHostToDevice));
    }
}

// GraphCache (GPU_Memory_Graph_Storage.cu:98-133): fragment of clique GPU i holds rows QT[r*Kg+i]
void GPUGraphStorage_GraphCache(GPUGraphStorage* g, int32_t* QT, int32_t Ki, int32_t Kg, int32_t capacity)
{
    if (!g || !QT || Kg < 1 || (Ki + 1) * Kg > g->partition_count) { LEGION_ARG_ERROR("GraphCache: bad clique"); return; }
    for (int i = 0; i < Kg; i++) {
        const int dev = Ki * Kg + i;
        if (is_remote_device(dev)) continue; // built by its own process, imported here over IPC
        DeviceGuard guard(dev);
        HIP_CHECK(hipDeviceSynchronize());
        free_fragment(g->frag[dev]);
        if (capacity <= 0) continue;
        GPUGraphStorage::Fragment& f = g->frag[dev];
        int64_t* neighbor_count = nullptr;
        int64_t* d_index = nullptr; // contiguous build copy of the fragment's indptr
        HIP_CHECK(hipMalloc(&neighbor_count, (size_t)capacity * sizeof(int64_t)));
        HIP_CHECK(hipMalloc(&d_index, ((size_t)capacity + 1) * sizeof(int64_t)));
        HIP_CHECK(hipMemset(d_index, 0, sizeof(int64_t)));
        launch_neighbor_count(nullptr, QT, Kg, i, capacity, g->node_num, g->csr_node_index_cpu, neighbor_count);
        inclusive_scan_i64(nullptr, neighbor_count, d_index + 1, capacity);
        int64_t total = 0;
        HIP_CHECK(hipMemcpy(&total, d_index + capacity, sizeof(int64_t), hipMemcpyDeviceToHost));
        HIP_CHECK(hipFree(neighbor_count));
        f.rows = capacity;
        f.edges = total;
        // indices chunks: chunk q ends where the last row that starts before its upper boundary ends
        const int nx = frag_ix_chunks(g, total);
        std::vector<int64_t> ends(nx, total);
        if (nx > 1) {
            int64_t* d_ends = nullptr;
            HIP_CHECK(hipMalloc(&d_ends, (size_t)nx * sizeof(int64_t)));
            launch_chunk_ends(nullptr, d_index, capacity, g->edge_shift, nx, d_ends);
            HIP_CHECK(hipMemcpy(ends.data(), d_ends, (size_t)(nx - 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
            HIP_CHECK(hipFree(d_ends));
        }
        f.ix.assign(nx, nullptr);
        for (int q = 0; q < nx; q++) {
            const int64_t elems = ends[q] - ((int64_t)q << g->edge_shift); // <= 0: no row starts in this chunk
            HIP_CHECK(hipMalloc(&f.ix[q], (size_t)(elems > 0 ? elems : 1) * sizeof(int32_t)));
        }
        int32_t** d_chunks = nullptr;
        HIP_CHECK(hipMalloc(&d_chunks, ((size_t)nx + 1) * sizeof(int32_t*))); // +1: see publish_fragment_tables
        HIP_CHECK(hipMemcpy(d_chunks, f.ix.data(), (size_t)nx * sizeof(int32_t*), hipMemcpyHostToDevice));
        launch_topo_fill_up(nullptr, QT, Kg, i, capacity, g->node_num, g->csr_node_index_cpu, g->csr_dst_node_ids_cpu, d_index, d_chunks, g->edge_shift);
        // indptr chunks (one entry of overlap); a single chunk adopts the build copy
        const int np = frag_ip_chunks(g, capacity);
        if (np == 1) {
            f.ip.assign(1, d_index);
            d_index = nullptr;
        } else {
            f.ip.assign(np, nullptr);
            const int64_t rpc = 1ll << g->row_shift;
            for (int q = 0; q < np; q++) {
                const int64_t r0 = q * rpc, n = std::min<int64_t>(rpc, capacity - r0) + 1;
                HIP_CHECK(hipMalloc(&f.ip[q], (size_t)n * sizeof(int64_t)));
                HIP_CHECK(hipMemcpy(f.ip[q], d_index + r0, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToDevice));
            }
        }
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipFree(d_chunks));
        if (d_index) HIP_CHECK(hipFree(d_index));
        for (auto* q : f.ip) LEGION_AUDIT_OWNER(q, dev, "GraphCache: indptr chunk of a fragment");
        for (auto* q : f.ix) LEGION_AUDIT_OWNER(q, dev, "GraphCache: indices chunk of a fragment");
        f.complete = true;
    }
    // every clique member sees every clique fragment (pointer tables copied D2D in the reference, :128-131)
    for (int i = 0; i < Kg; i++)
        for (int j = 0; j < Kg; j++) g->view[Ki * Kg + i][Ki * Kg + j] = true;
    publish_fragment_tables(g);
}

int64_t GPUGraphStorage_ReplicateToDevices(GPUGraphStorage* g)
{
    if (!g || !g->csr_node_index_cpu || g->csr_location == LEGION_LOC_DEVICE) return 0; // already HBM resident
    replicate_table<int64_t>(g->csr_node_index_cpu, (int64_t)g->node_num + 1, g->partition_count, g->replica_indptr);
    replicate_table<int32_t>(g->csr_dst_node_ids_cpu, g->edge_num, g->partition_count, g->replica_indices);
    return ((int64_t)g->node_num + 1) * 8 + g->edge_num * 4;
}

void GPUGraphStorage_Finalize(GPUGraphStorage* g)
{
    if (!g) return;
    free_replicas(g->replica_indptr);
    free_replicas(g->replica_indices);
    for (size_t i = 0; i < g->frag.size(); i++) {
        if (!is_remote_device((int)i) || g->frag[i].imported) { DeviceGuard guard((int)i); free_fragment(g->frag[i]); }
        if (g->d_frag_tab[i]) { DeviceGuard guard((int)i); (void)hipFree(g->d_frag_tab[i]); g->d_frag_tab[i] = nullptr; }
    }
    if (g->owns_csr) {
        host_free_space(g->csr_node_index_cpu);
        host_free_space(g->csr_dst_node_ids_cpu);
        g->owns_csr = false;
    }
}
int32_t GPUGraphStorage_GetPartitionCount(const GPUGraphStorage* g) { return g->partition_count; }
int64_t* GPUGraphStorage_GetCSRNodeIndexCPU(const GPUGraphStorage* g) { return g->csr_node_index_cpu; }
int32_t* GPUGraphStorage_GetCSRNodeMatrixCPU(const GPUGraphStorage* g) { return g->csr_dst_node_ids_cpu; }
static bool frag_args_ok(const GPUGraphStorage* g, int32_t dev_id, int32_t part_id)
{
    return g && dev_id >= 0 && dev_id < g->partition_count && part_id >= 0 && part_id < g->partition_count;
}
int64_t* GPUGraphStorage_GetFragmentIndex(const GPUGraphStorage* g, int32_t dev_id, int32_t part_id)
{   // first chunk (the whole indptr when the fragment has one chunk)
    if (!frag_args_ok(g, dev_id, part_id) || !g->view[dev_id][part_id] || g->frag[part_id].ip.empty()) return nullptr;
    return g->frag[part_id].ip[0];
}
int32_t* GPUGraphStorage_GetFragmentMatrix(const GPUGraphStorage* g, int32_t dev_id, int32_t part_id)
{
    if (!frag_args_ok(g, dev_id, part_id) || !g->view[dev_id][part_id] || g->frag[part_id].ix.empty()) return nullptr;
    return g->frag[part_id].ix[0];
}
int32_t GPUGraphStorage_FragmentRows(const GPUGraphStorage* g, int32_t dev_id) { return frag_args_ok(g, dev_id, 0) ? g->frag[dev_id].rows : 0; }
int64_t GPUGraphStorage_FragmentEdges(const GPUGraphStorage* g, int32_t dev_id) { return frag_args_ok(g, dev_id, 0) ? g->frag[dev_id].edges : 0; }
int32_t GPUGraphStorage_FragmentChunkCount(const GPUGraphStorage* g, int32_t dev_id, int32_t which)
{
    if (!frag_args_ok(g, dev_id, 0)) return 0;
    return which == 0 ? (int32_t)g->frag[dev_id].ip.size() : (int32_t)g->frag[dev_id].ix.size();
}
int64_t GPUGraphStorage_FragmentChunkSpan(const GPUGraphStorage* g, int32_t which)
{
    return g ? (1ll << (which == 0 ? g->row_shift : g->edge_shift)) : 0;
}
void* GPUGraphStorage_GetFragmentChunk(const GPUGraphStorage* g, int32_t dev_id, int32_t which, int32_t chunk)
{
    if (!frag_args_ok(g, dev_id, 0) || chunk < 0 || chunk >= GPUGraphStorage_FragmentChunkCount(g, dev_id, which)) return nullptr;
    return which == 0 ? (void*)g->frag[dev_id].ip[chunk] : (void*)g->frag[dev_id].ix[chunk];
}
int GPUGraphStorage_ExportFragmentChunk(GPUGraphStorage* g, int32_t dev_id, int32_t which, int32_t chunk, void* handle64)
{
    void* p = GPUGraphStorage_GetFragmentChunk(g, dev_id, which, chunk);
    if (!p || !handle64 || g->frag[dev_id].imported) { LEGION_ARG_ERROR("ExportFragmentChunk: no such local chunk"); return -1; }
    DeviceGuard guard(dev_id);
    if (!ipc_export_ok(p, "ExportFragmentChunk")) return -1;
    HIP_CHECK(hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, p));
    return error_pending() ? -1 : 0;
}
int GPUGraphStorage_ImportFragmentChunk(GPUGraphStorage* g, int32_t owner_dev, int32_t viewer_dev, int32_t which, int32_t chunk,
                                        const void* handle64, int32_t rows, int64_t edges)
{
    if (!frag_args_ok(g, owner_dev, viewer_dev) || !handle64 || !is_remote_device(owner_dev) || rows <= 0 || edges < 0) { LEGION_ARG_ERROR("ImportFragmentChunk: owner must be a remote member"); return -1; }
    GPUGraphStorage::Fragment& f = g->frag[owner_dev];
    if (!f.imported) {
        f.rows = rows; f.edges = edges; f.imported = true;
        f.ip.assign(frag_ip_chunks(g, rows), nullptr);
        f.ix.assign(frag_ix_chunks(g, edges), nullptr);
    }
    if (f.rows != rows || f.edges != edges) { LEGION_ARG_ERROR("ImportFragmentChunk: rows/edges differ from the first chunk's"); return -1; }
    const int n = which == 0 ? (int)f.ip.size() : (int)f.ix.size();
    if (chunk < 0 || chunk >= n) { LEGION_ARG_ERROR("ImportFragmentChunk: chunk out of range"); return -1; }
    void*& slot = which == 0 ? (void*&)f.ip[chunk] : (void*&)f.ix[chunk];
    {   // lower bound of what the exporter allocated for this chunk
        const int64_t bytes = which == 0 ? (std::min<int64_t>(rows, 1ll << g->row_shift) + 1) * (int64_t)sizeof(int64_t)
                                         : std::min<int64_t>(edges, 1ll << g->edge_shift) * (int64_t)sizeof(int32_t);
        if (!slot && !ipc_size_ok(bytes, "ImportFragmentChunk")) return -1;
    }
    if (!slot) {
        DeviceGuard guard(viewer_dev);
        hipIpcMemHandle_t h;
        memcpy(&h, handle64, sizeof(h));
        void* p = nullptr;
        HIP_CHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        if (!p) return -1;
        slot = p;
    }
    g->view[viewer_dev][owner_dev] = true;
    f.complete = fragment_complete(g, owner_dev);
    if (f.complete) publish_fragment_tables(g);
    return 0;
}
int GPUGraphStorage_ExportFragment(GPUGraphStorage* g, int32_t dev_id, void* handle_indptr64, void* handle_indices64, int32_t* rows_out)
{   // single-chunk fragments only; chunked fragments use the *Chunk calls
    if (GPUGraphStorage_FragmentChunkCount(g, dev_id, 0) != 1 || GPUGraphStorage_FragmentChunkCount(g, dev_id, 1) != 1) { LEGION_ARG_ERROR("ExportFragment: fragment has several chunks, use ExportFragmentChunk"); return -1; }
    if (rows_out) *rows_out = g->frag[dev_id].rows;
    if (GPUGraphStorage_ExportFragmentChunk(g, dev_id, 0, 0, handle_indptr64) != 0) return -1;
    return GPUGraphStorage_ExportFragmentChunk(g, dev_id, 1, 0, handle_indices64);
}
int GPUGraphStorage_ImportFragment(GPUGraphStorage* g, int32_t owner_dev, int32_t viewer_dev, const void* handle_indptr64,
                                   const void* handle_indices64, int32_t rows)
{   // single-chunk form: the edge count is not known here, any value inside the first chunk selects one chunk
    if (GPUGraphStorage_ImportFragmentChunk(g, owner_dev, viewer_dev, 0, 0, handle_indptr64, rows, 1) != 0) return -1;
    return GPUGraphStorage_ImportFragmentChunk(g, owner_dev, viewer_dev, 1, 0, handle_indices64, rows, 1);
}
void GPUGraphStorage_Delete(GPUGraphStorage* g)
{
    if (!g) return;
    GPUGraphStorage_Finalize(g);
    delete g;
}

// ================================= node storage =====================================================
GPUNodeStorage* NewGPUMemoryNodeStorage(void) { return new GPUNodeStorage(); }

static int32_t* upload_i32(const int32_t* src, int32_t n)
{
    int32_t* d = nullptr;
    HIP_CHECK(hipMalloc(&d, (size_t)(n > 0 ? n : 1) * sizeof(int32_t)));
    if (n > 0 && src) HIP_CHECK(hipMemcpy(d, src, (size_t)n * sizeof(int32_t), hipMemcpyDefault));
    return d;
}

void GPUNodeStorage_Build(GPUNodeStorage* n, const LegionBuildInfo* info)
{
    if (!n || !info) { LEGION_ARG_ERROR("GPUNodeStorage_Build: null argument"); return; }
    const int P = info->partition_count;
    if (P < 1 || P > kMaxParts) { LEGION_ARG_ERROR("GPUNodeStorage_Build: partition_count must be 1..8"); return; }
    n->partition_count = P;
    n->total_num_nodes = info->total_num_nodes;
    n->float_attr_len = info->float_attr_len;
    // float_attr_pitch is an extension field behind the reference's BuildInfo: 0 = dense.  Anything else must describe a
    // layout the gather can read: at least F floats, and -- when F allows 16-byte chunks -- rows that stay 16-byte aligned
    // (the float4 path is chosen from F and the base pointers).  A caller built against the shorter struct must zero it.
    if (info->float_attr_pitch != 0 && (info->float_attr_pitch < info->float_attr_len ||
                                        (info->float_attr_len % 4 == 0 && info->float_attr_pitch % 4 != 0))) {
        LEGION_ARG_ERROR("GPUNodeStorage_Build: float_attr_pitch must be 0 (dense) or >= float_attr_len, and a multiple of 4 floats when float_attr_len is");
        return;
    }
    n->float_attr_pitch = info->float_attr_pitch > info->float_attr_len ? info->float_attr_pitch : info->float_attr_len;
    n->replica_pitch = 0;
    n->features_location = info->features_location;
    bool owns = false;
    n->float_attrs = info->host_float_attrs
        ? adopt_table<float>(info->host_float_attrs, (int64_t)info->total_num_nodes * n->float_attr_pitch, info->features_location, &owns)
        : nullptr;
    n->owns_features = owns;
    n->replica_attrs.assign(P, nullptr);
    n->training_set_num.assign(P, 0); n->validation_set_num.assign(P, 0); n->testing_set_num.assign(P, 0);
    n->training_set_ids.assign(P, nullptr); n->validation_set_ids.assign(P, nullptr); n->testing_set_ids.assign(P, nullptr);
    n->training_labels.assign(P, nullptr); n->validation_labels.assign(P, nullptr); n->testing_labels.assign(P, nullptr);
    for (int p = 0; p < P; p++) { // GPU_Memory_Node_Storage.cu:41-96
        if (is_remote_device(p)) continue; // that partition's seed sets live in its own process
        DeviceGuard guard(p);
        if (info->training_set_num) {
            n->training_set_num[p] = info->training_set_num[p];
            n->training_set_ids[p] = upload_i32(info->training_set_ids[p], info->training_set_num[p]);
            n->training_labels[p] = upload_i32(info->training_labels[p], info->training_set_num[p]);
        }
        if (info->validation_set_num) {
            n->validation_set_num[p] = info->validation_set_num[p];
            n->validation_set_ids[p] = upload_i32(info->validation_set_ids[p], info->validation_set_num[p]);
            n->validation_labels[p] = upload_i32(info->validation_labels[p], info->validation_set_num[p]);
        }
        if (info->testing_set_num) {
            n->testing_set_num[p] = info->testing_set_num[p];
            n->testing_set_ids[p] = upload_i32(info->testing_set_ids[p], info->testing_set_num[p]);
            n->testing_labels[p] = upload_i32(info->testing_labels[p], info->testing_set_num[p]);
        }
        LEGION_AUDIT_OWNER(n->training_set_ids[p], p, "GPUNodeStorage_Build: seed list");
    }
}

int64_t GPUNodeStorage_ReplicateToDevices(GPUNodeStorage* n)
{
    if (!n || !n->float_attrs || n->features_location == LEGION_LOC_DEVICE) return 0;
    const int32_t F = n->float_attr_len, pitch = legion_row_pitch(F);
    const int64_t V = n->total_num_nodes;
    n->replica_pitch = pitch;
    if (pitch == n->float_attr_pitch) {
        replicate_table<float>(n->float_attrs, V * pitch, n->partition_count, n->replica_attrs);
        return V * pitch * 4;
    }
    // the replica gets a line-aligned row pitch (legion_row_pitch): one pitched copy per distinct physical device
    std::vector<std::pair<int, float*>> per_phys;
    for (int p = 0; p < n->partition_count; p++) {
        if (is_remote_device(p) || n->replica_attrs[p]) continue;
        const int phys = physical_device(p);
        float* have = nullptr;
        for (auto& e : per_phys) if (e.first == phys) have = e.second;
        if (!have) {
            DeviceGuard guard(p);
            HIP_CHECK(hipMalloc(&have, (size_t)V * pitch * sizeof(float)));
            if (have) {
                HIP_CHECK(hipMemset(have, 0, (size_t)V * pitch * sizeof(float)));
                HIP_CHECK(hipMemcpy2D(have, (size_t)pitch * sizeof(float), n->float_attrs, (size_t)n->float_attr_pitch * sizeof(float),
                                      (size_t)F * sizeof(float), (size_t)V, hipMemcpyDefault));
            }
            per_phys.emplace_back(phys, have);
        }
        LEGION_AUDIT_SHARE(have, p);
        n->replica_attrs[p] = have;
    }
    return V * pitch * 4;
}

void GPUNodeStorage_Finalize(GPUNodeStorage* n)
{
    if (!n) return;
    free_replicas(n->replica_attrs);
    auto drop = [](std::vector<int32_t*>& v) { for (auto& p : v) { if (p) (void)hipFree(p); p = nullptr; } };
    drop(n->training_set_ids); drop(n->validation_set_ids); drop(n->testing_set_ids);
    drop(n->training_labels); drop(n->validation_labels); drop(n->testing_labels);
    if (n->owns_features) { host_free_space(n->float_attrs); n->owns_features = false; }
}
#define NS_GETTER(name, field) \
    int32_t* GPUNodeStorage_##name(const GPUNodeStorage* n, int32_t part_id) { \
        return (part_id >= 0 && part_id < n->partition_count) ? n->field[part_id] : nullptr; }
NS_GETTER(GetTrainingSetIds, training_set_ids)
NS_GETTER(GetValidationSetIds, validation_set_ids)
NS_GETTER(GetTestingSetIds, testing_set_ids)
NS_GETTER(GetTrainingLabels, training_labels)
NS_GETTER(GetValidationLabels, validation_labels)
NS_GETTER(GetTestingLabels, testing_labels)
#undef NS_GETTER
int32_t GPUNodeStorage_TrainingSetSize(const GPUNodeStorage* n, int32_t p) { return (p >= 0 && p < n->partition_count) ? n->training_set_num[p] : 0; }
int32_t GPUNodeStorage_ValidationSetSize(const GPUNodeStorage* n, int32_t p) { return (p >= 0 && p < n->partition_count) ? n->validation_set_num[p] : 0; }
int32_t GPUNodeStorage_TestingSetSize(const GPUNodeStorage* n, int32_t p) { return (p >= 0 && p < n->partition_count) ? n->testing_set_num[p] : 0; }
int32_t GPUNodeStorage_TotalNodeNum(const GPUNodeStorage* n) { return n->total_num_nodes; }
float* GPUNodeStorage_GetAllFloatAttr(const GPUNodeStorage* n) { return n->float_attrs; }
int32_t GPUNodeStorage_GetFloatAttrLen(const GPUNodeStorage* n) { return n->float_attr_len; }
void GPUNodeStorage_Delete(GPUNodeStorage* n)
{
    if (!n) return;
    GPUNodeStorage_Finalize(n);
    delete n;
}

} // extern "C"

// ================================= memory pool ======================================================
GPUMemoryPool::GPUMemoryPool(int32_t depth)
{
    pipeline_depth = depth > 0 ? depth : 1;
    float_features.assign(pipeline_depth, nullptr);
    labels.assign(pipeline_depth, nullptr);
    node_counter.assign(pipeline_depth, nullptr);
    edge_counter.assign(pipeline_depth, nullptr);
    sampled_ids.assign(pipeline_depth, nullptr);
    agg_src_off.assign(pipeline_depth, nullptr);
    agg_dst_off.assign(pipeline_depth, nullptr);
}

extern "C" {

GPUMemoryPool* NewGPUMemoryPool(int32_t pipeline_depth) { return new GPUMemoryPool(pipeline_depth); }

// Server.cu:184-196 (num_ids_) and :216-231 (scratch), sized for H hops
void GPUMemoryPool_AllocateScratch(GPUMemoryPool* p, int32_t total_num_nodes, int32_t batch_size,
                                   const int32_t* fanout, int32_t hops)
{
    if (!p || hops < 1 || hops > LEGION_MAX_HOPS || batch_size < 1 || total_num_nodes < 1) { LEGION_ARG_ERROR("GPUMemoryPool_AllocateScratch: bad arguments"); return; }
    if (p->owns_scratch) GPUMemoryPool_Finalize(p); // re-sizing an initialised pool: release the old scratch first
    p->V = total_num_nodes; p->batch_size = batch_size; p->hops = hops;
    int64_t ids = batch_size, cur = batch_size, max_slots = 0;
    p->level_bound[0] = batch_size;
    for (int h = 0; h < hops; h++) {
        p->fanout[h] = fanout[h];
        cur *= fanout[h];
        if (cur > max_slots) max_slots = cur;
        ids += cur;
        if (ids >= (1ll << 31)) { LEGION_ARG_ERROR("GPUMemoryPool_AllocateScratch: batch*fanouts exceeds int32"); return; }
        if (cur >= (1ll << 30)) { LEGION_ARG_ERROR("GPUMemoryPool_AllocateScratch: a hop of 2^30 or more slots exceeds the slot-state encoding"); return; }
        p->level_bound[h + 1] = (int32_t)cur;
    }
    p->num_ids = (int32_t)ids;
    p->max_slots = (int32_t)max_slots;
    p->max_tiles = (int32_t)((max_slots + kTile - 1) / kTile);
    p->owns_scratch = true;
    HIP_CHECK(hipMalloc(&p->pos_map, (size_t)total_num_nodes * sizeof(pos_t)));
    HIP_CHECK(hipMemset(p->pos_map, 0xFF, (size_t)total_num_nodes * sizeof(pos_t)));
    p->batch_serial = 0;
    HIP_CHECK(hipMalloc(&p->ctl, sizeof(BatchCtl)));
    { const BatchCtl c{0, kEpochTop}; HIP_CHECK(hipMemcpy(p->ctl, &c, sizeof(c), hipMemcpyHostToDevice)); }
    p->ctl_synced = false;
    HIP_CHECK(hipHostMalloc((void**)&p->rows_seen, (LEGION_MAX_HOPS + 2) * sizeof(int32_t), hipHostMallocMapped));
    memset(p->rows_seen, 0, (LEGION_MAX_HOPS + 2) * sizeof(int32_t));
    HIP_CHECK(hipHostGetDevicePointer((void**)&p->rows_seen_dev, p->rows_seen, 0));
    HIP_CHECK(hipMalloc(&p->cand, (size_t)p->max_slots * sizeof(int32_t)));
    for (auto& a : p->aux2) HIP_CHECK(hipMalloc(&a, (size_t)p->max_slots * sizeof(int32_t)));
    HIP_CHECK(hipMalloc(&p->tile_edge, (size_t)(p->max_tiles + 1) * sizeof(int32_t)));
    HIP_CHECK(hipMalloc(&p->tile_node, (size_t)(p->max_tiles + 1) * sizeof(int32_t)));
    HIP_CHECK(hipMalloc(&p->tile_pre, (size_t)(p->max_tiles + 1) * sizeof(int2)));
    HIP_CHECK(hipMalloc(&p->chunk_tot, (size_t)kMaxChunks * sizeof(int2)));
    HIP_CHECK(hipMalloc(&p->hop_state, sizeof(HopState)));
    HIP_CHECK(hipMalloc(&p->cache_search_buffer, (size_t)p->num_ids * sizeof(int32_t)));
    HIP_CHECK(hipMalloc((void**)&p->row_ptr, (size_t)p->num_ids * sizeof(float*)));
    HIP_CHECK(hipMalloc(&p->agg_src_ids, (size_t)p->num_ids * sizeof(int32_t)));
    HIP_CHECK(hipMalloc(&p->tmp_part_ind, (size_t)p->num_ids));
    HIP_CHECK(hipMalloc(&p->tmp_part_off, (size_t)p->num_ids * sizeof(int32_t)));
    HIP_CHECK(hipDeviceSynchronize());
}
int32_t GPUMemoryPool_NumIds(const GPUMemoryPool* p) { return p->num_ids; }

#define POOL_PIPE_SETTER(name, field, type) \
    void GPUMemoryPool_Set##name(GPUMemoryPool* p, type* ptr, int32_t pipe) { \
        if (pipe < 0 || pipe >= p->pipeline_depth) { LEGION_ARG_ERROR("GPUMemoryPool_Set" #name ": bad pipe"); return; } \
        p->field[pipe] = ptr; } \
    type* GPUMemoryPool_Get##name(const GPUMemoryPool* p) { return p->field[p->current_pipe]; }
POOL_PIPE_SETTER(SampledIds, sampled_ids, int32_t)
POOL_PIPE_SETTER(FloatFeatures, float_features, float)
POOL_PIPE_SETTER(Labels, labels, int32_t)
POOL_PIPE_SETTER(AggSrcOf, agg_src_off, int32_t)
POOL_PIPE_SETTER(AggDstOf, agg_dst_off, int32_t)
POOL_PIPE_SETTER(NodeCounter, node_counter, int32_t)
POOL_PIPE_SETTER(EdgeCounter, edge_counter, int32_t)
#undef POOL_PIPE_SETTER
void GPUMemoryPool_SetFeatureRows(GPUMemoryPool* p, int32_t rows) { p->feature_rows = rows; }
void GPUMemoryPool_SetCurrentPipe(GPUMemoryPool* p, int32_t pipe) { p->current_pipe = pipe % p->pipeline_depth; }
void GPUMemoryPool_SetCurrentMode(GPUMemoryPool* p, int32_t mode) { p->mode = mode; }
void GPUMemoryPool_SetIter(GPUMemoryPool* p, int32_t iter) { p->iter = iter; }
int32_t GPUMemoryPool_GetCurrentMode(const GPUMemoryPool* p) { return p->mode; }
int32_t GPUMemoryPool_GetIter(const GPUMemoryPool* p) { return p->iter; }
int32_t* GPUMemoryPool_GetAggSrcId(const GPUMemoryPool* p) { return p->agg_src_ids; }
int32_t* GPUMemoryPool_GetCacheSearchBuffer(const GPUMemoryPool* p) { return p->cache_search_buffer; }
char* GPUMemoryPool_GetTmpPartIdx(const GPUMemoryPool* p) { return (char*)p->tmp_part_ind; }
int32_t* GPUMemoryPool_GetTmpPartOff(const GPUMemoryPool* p) { return p->tmp_part_off; }
uint64_t* GPUMemoryPool_GetPositionMap(const GPUMemoryPool* p) { return (uint64_t*)p->pos_map; }
int32_t* GPUMemoryPool_GetCandidateBuffer(const GPUMemoryPool* p) { return p->cand; }
uint32_t GPUMemoryPool_GetBatchSerial(const GPUMemoryPool* p) { return p->batch_serial; }
void GPUMemoryPool_SetBatchSerial(GPUMemoryPool* p, uint32_t serial) { p->batch_serial = serial; p->ctl_synced = false; }

void GPUMemoryPool_Finalize(GPUMemoryPool* p)
{
    if (!p || !p->owns_scratch) return;
    GPUMemoryPool_ReleasePeerExchange(p);
    (void)hipFree(p->pos_map); (void)hipFree(p->cand); for (auto& a : p->aux2) { (void)hipFree(a); a = nullptr; } (void)hipFree(p->tile_edge); (void)hipFree(p->tile_node); (void)hipFree(p->tile_pre); (void)hipFree(p->chunk_tot); p->tile_pre = p->chunk_tot = nullptr;
    (void)hipFree(p->hop_state); (void)hipFree(p->cache_search_buffer); (void)hipFree((void*)p->row_ptr); p->row_ptr = nullptr; (void)hipFree(p->agg_src_ids);
    (void)hipFree(p->tmp_part_ind); (void)hipFree(p->tmp_part_off); (void)hipFree(p->ctl); p->ctl = nullptr; if (p->rows_seen) { (void)hipHostFree(p->rows_seen); p->rows_seen = nullptr; p->rows_seen_dev = nullptr; }
    p->pos_map = nullptr; p->cand = nullptr; p->tile_edge = p->tile_node = nullptr; p->hop_state = nullptr;
    p->cache_search_buffer = p->agg_src_ids = p->tmp_part_off = nullptr; p->tmp_part_ind = nullptr;
    p->owns_scratch = false;
}
void GPUMemoryPool_Delete(GPUMemoryPool* p)
{
    if (!p) return;
    GPUMemoryPool_Finalize(p);
    delete p;
}

} // extern "C"
