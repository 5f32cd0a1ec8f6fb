// ipc_service -- the trainer-side Python module of the hand-off (drop-in for
// pytorch_extension/ipc_service.cpp:14-93 + ipc_cuda_kernel.cu:178-230 of the reference):
//   initialize() get_next(feature_dim) get_block_size() get_steps() synchronize() finalize()
// All device work goes through the C ABI of liblegion_amd.so (legion_ipc_client_*); tensors are
// zero-copy torch::from_blob views of server-owned device memory, valid until synchronize().
// For 2 hops get_next returns the reference's 7 tensors
//   [ids, features, labels, b1_src, b1_dst, b2_src, b2_dst]   (b2_* alias the prefix of b1_*);
// for H hops it returns 3 + 2H tensors, block k covering the edges of hops 1..H-k+1.
#include <torch/extension.h>

#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/legion_amd.h"

static LegionIPCClient* env = nullptr;
static int32_t h_node_counter[16];
static int32_t h_edge_counter[16];
static int32_t g_hops = 2;
// ONE consumer thread per process is the contract (the reference's trainer loop, legion_graphsage.py:72-89):
// get_next / synchronize run without the GIL and share `env`, the two counter arrays and the client's current pipe, so the entry points
// are serialised by this lock -- uncontended in the reference's loop, and a second Python thread gets whole counters instead of torn ones.
static std::mutex g_mu;

static void require_env()
{
    TORCH_CHECK(env != nullptr, "ipc_service.initialize() was not called");
}

void InitializeIPC()
{
    env = legion_ipc_client_open(-1); // current device == torch.cuda.set_device(rank) (ipc_cuda_kernel.cu:41)
    TORCH_CHECK(env != nullptr, "ipc_service: cannot attach to the sampling server: ", legion_last_error());
    g_hops = legion_ipc_client_hops(env);
}

void FinalizeIPC()
{
    if (env) legion_ipc_client_close(env);
    env = nullptr;
}

std::vector<torch::Tensor> get_next(int feature_dim)
{
    std::lock_guard<std::mutex> lock(g_mu);
    require_env();
    legion_ipc_client_wait(env); // env->Wait(), ipc_service.cpp:42
    legion_ipc_client_read_counters(env, h_node_counter, h_edge_counter);
    // a server that failed mid-batch posts the pipe with every node-counter word at -1 (runner.cpp, post_poisoned)
    TORCH_CHECK(h_node_counter[0] >= 0, "ipc_service: the sampling server failed (poisoned batch posted)");
    const int dev = GetGPUDevice();
    const auto device = torch::Device(torch::kCUDA, dev);
    const auto i32 = torch::TensorOptions().dtype(torch::kI32).device(device);
    const auto f32 = torch::TensorOptions().dtype(torch::kF32).device(device);
    const int H = g_hops;
    const int64_t n_nodes = h_node_counter[5 + 2 * H];
    // the feature buffer holds a bounded number of rows (1.2 x the largest pre-sampled batch, Server.cu:275): a batch that reaches more
    // nodes must not be viewed as [n, F] (the reference does, unchecked: ipc_cuda_kernel.cu:200 -- a read past the allocation)
    const int64_t rows = legion_ipc_client_feature_rows(env);
    TORCH_CHECK(rows <= 0 || n_nodes <= rows, "ipc_service: the batch has ", n_nodes, " nodes but the server's feature buffer holds ", rows,
                " rows (sized from the pre-sampling epoch; use a training batch size >= the validation / test batch size)");
    for (int w = 0; w <= 4; w++)    // legion_ipc_client_open refuses a server that has not registered its buffers; never build a tensor on a null one
        TORCH_CHECK(legion_ipc_client_buffer(env, w) != nullptr, "ipc_service: hand-off buffer ", w, " of this GPU was never registered by the server");
    std::vector<torch::Tensor> out;
    out.push_back(torch::from_blob(legion_ipc_client_buffer(env, 0), {n_nodes}, i32));
    out.push_back(torch::from_blob(legion_ipc_client_buffer(env, 1), {n_nodes, (int64_t)feature_dim}, f32));
    out.push_back(torch::from_blob(legion_ipc_client_buffer(env, 2), {(int64_t)h_node_counter[5]}, i32));
    for (int k = 1; k <= H; k++) {
        const int64_t n_edges = h_edge_counter[2 + (H - k + 1)]; // ec[4], ec[3] at H = 2 (ipc_cuda_kernel.cu:198-213)
        out.push_back(torch::from_blob(legion_ipc_client_buffer(env, 3), {n_edges}, i32));
        out.push_back(torch::from_blob(legion_ipc_client_buffer(env, 4), {n_edges}, i32));
    }
    return out;
}

// [b1_src_nodes, b1_dst_nodes, b2_src_nodes, b2_dst_nodes, ...] = [nc9, nc7, nc7, nc5] at H = 2
// (ipc_service.cpp:60-72)
std::vector<int> get_block_size()
{
    std::lock_guard<std::mutex> lock(g_mu);
    std::vector<int> ret;
    const int H = g_hops;
    for (int k = 1; k <= H; k++) {
        ret.push_back(h_node_counter[5 + 2 * (H - k + 1)]);
        ret.push_back(h_node_counter[5 + 2 * (H - k)]);
    }
    return ret;
}

std::vector<int32_t> get_steps()
{
    require_env();
    int32_t s[3];
    legion_ipc_client_steps(env, s);
    return {s[0], s[1], s[2]};
}

void Synchronize()
{
    std::lock_guard<std::mutex> lock(g_mu);
    require_env();
    // env->Post(), ipc_service.cpp:83-85.  legion_ipc_client_post waits for the device first: the reference trainers call this with
    // their optimizer step still queued (legion_graphsage.py:93-116), and a posted pipe is overwritten by the server
    legion_ipc_client_post(env);
}

int get_hops() { return g_hops; }

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    // get_next blocks on the pipe's semaphore and synchronize on the trainer's device: neither touches Python state, so both run without the GIL
    // (the reference holds it: a trainer's other Python threads stall for as long as the server takes to produce a batch)
    m.def("get_next", &get_next, "dataset get next (HIP)", pybind11::call_guard<pybind11::gil_scoped_release>());
    m.def("get_block_size", &get_block_size, "get dgl block size");
    m.def("get_steps", &get_steps, "get steps");
    m.def("initialize", &InitializeIPC, "InitializeIPC");
    m.def("finalize", &FinalizeIPC, "FinalizeIPC");
    m.def("synchronize", &Synchronize, "synchronize", pybind11::call_guard<pybind11::gil_scoped_release>());
    m.def("get_hops", &get_hops, "number of hops the server samples (extension)");
}
