// batch_graph.cpp -- one mini-batch (S1..S5: seed, H x sampler, gathers, planner) recorded once as a
// hipGraph and replayed with a single launch per batch.
//
// The reference drives every batch from the host: ~(2H+4) operator launches, each with per-batch
// arguments (Server.cu:301-328).  Here the two values that change from batch to batch -- the batch
// cursor inside the seed list and the position-table epoch -- live in device memory (BatchCtl):
// k_seed<SELF> reads them, k_advance (the last node of the graph) steps them, so the recorded graph
// has no per-batch arguments at all.  The host keeps a mirror (batch_serial / ctl_counter) and only
// launches k_set_cursor when the next batch is not the successor of the previous graph launch
// (first batch, epoch / mode change, or after a host-driven batch on the same pool).
#include "internal.h"

#include "audit_hooks.h"

using namespace legion;

struct LegionBatchGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    GPUMemoryPool* pool = nullptr;
};

extern "C" {

int GPUMemoryPool_BeginBatchCapture(GPUMemoryPool* p, void* stream)
{
    if (!p || !p->owns_scratch || !p->ctl) { LEGION_ARG_ERROR("BeginBatchCapture: GPUMemoryPool_AllocateScratch was not called"); return -1; }
    if (p->capturing) { LEGION_ARG_ERROR("BeginBatchCapture: a capture is already running on this pool"); return -1; }
    if (!stream) { LEGION_ARG_ERROR("BeginBatchCapture: the legacy null stream cannot be captured"); return -1; }
    warm_static_tables(); // no allocation / copy may happen between Begin and End
    HIP_CHECK(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    if (error_pending()) return -1;
    p->capturing = true;
    return 0;
}

LegionBatchGraph* GPUMemoryPool_EndBatchCapture(GPUMemoryPool* p, void* stream)
{
    if (!p || !p->capturing) { LEGION_ARG_ERROR("EndBatchCapture: no capture is running on this pool"); return nullptr; }
    launch_advance((hipStream_t)stream, p->ctl);
    p->capturing = false;
    LegionBatchGraph* g = new LegionBatchGraph();
    g->pool = p;
    HIP_CHECK(hipStreamEndCapture((hipStream_t)stream, &g->graph));
    if (g->graph) HIP_CHECK(hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0));
    if (!g->exec || error_pending()) {
        if (g->graph) (void)hipGraphDestroy(g->graph);
        delete g;
        return nullptr;
    }
    return g;
}

// Run the recorded batch for seed-list position `counter` (what batch_generator_kernel's `counter` is).
int LegionBatchGraph_Launch(LegionBatchGraph* g, void* stream, int32_t counter)
{
    if (!g || !g->exec || !g->pool || !g->pool->ctl) { LEGION_ARG_ERROR("LegionBatchGraph_Launch: null graph"); return -1; }
    GPUMemoryPool* p = g->pool;
    if (p->capturing) { LEGION_ARG_ERROR("LegionBatchGraph_Launch: pool is being captured"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (++p->batch_serial >= kSerialLimit) { // epoch space exhausted: wipe once and start over (as batch_generator_kernel)
        HIP_CHECK(hipMemsetAsync(p->pos_map, 0xFF, (size_t)p->V * sizeof(pos_t), s));
        p->batch_serial = 1;
        p->ctl_synced = false;
    }
    if (!p->ctl_synced || p->ctl_counter != counter) launch_set_cursor(s, p->ctl, counter, kEpochTop - p->batch_serial);
    HIP_CHECK(hipGraphLaunch(g->exec, s));
    p->ctl_synced = true;     // k_advance left (counter + 1, next epoch) in ctl
    p->ctl_counter = counter + 1;
    return error_pending() ? -1 : 0;
}

void LegionBatchGraph_Delete(LegionBatchGraph* g)
{
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

} // extern "C"
