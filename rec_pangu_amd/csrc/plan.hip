// LAUNCH PLANS: the library's own replayable form of a sequence of its kernel launches (host code only).
//
// Why not a hipGraph alone: on this runtime a replayed graph costs the device ~10 us per kernel node more than the same
// launches issued on a stream (B = 65536: 1.5 - 1.6 ms replayed against 1.26 ms eager, profiles/microbench/probes/
// probe_graph5.py), so a captured step only paid off for small, host-bound batches — while the eager step needs ~1 ms
// of Python / ctypes / autograd host time per step.  A plan keeps what the capture gives (one host call per step, frozen
// arguments, device-resident step counters) and re-issues the launches with hipLaunchKernel: the device sees exactly the
// eager stream of kernels, the host spends a few microseconds per launch.
//
// Recording: between rp_plan_begin and rp_plan_end every launch of the library (common.h: rp_launch) is appended with its
// packed arguments — from whatever host thread issues it (the autograd engine runs backward nodes on its own thread).
// rec_pangu_amd/graph_step.py records while a stream capture is active: the capture supplies the allocator's private
// pool (the addresses baked into the plan stay reserved) and, through rp_graph_node_counts, the proof that the step holds
// no launch the plan has not seen (an ATen kernel, a memset or memcpy node): such a step is replayed as a hipGraph instead.
//
// Sections: launches recorded while rp_plan_section(1) is in force are independent of the rest of the plan (the row sort
// of the NEXT batch) and are re-issued on the plan's side stream, forked at the start of the replay and joined at its end —
// the overlap the eager path gets from its side stream, which a fork inside a hipGraph does not give on this runtime.
#include "common.h"
#include <cxxabi.h>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <atomic>
#include <mutex>
#include <vector>

namespace {

struct PlanNode {
    const void *func;
    dim3 grid, block;
    unsigned shmem;
    int section;
    hipStream_t rec_stream;
    size_t blob_at;                 // offset of the packed arguments in Plan::blob
    std::vector<size_t> offs;       // argument offsets inside the packed buffer
};

struct Plan {
    std::vector<PlanNode> nodes;
    std::vector<char> blob;         // every node's packed arguments, 16-byte aligned each
    std::vector<void *> ptrs;       // per node: pointers into blob (built by rp_plan_end)
    std::vector<size_t> ptrs_at;
    int section = 0;
    int n_side = 0;
    size_t fork_at = 0;             // the side section is forked in front of main-section node `fork_at` (node index)
    bool ended = false;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int n_inline = 0;               // launches of section 2 (forked where they were recorded, joined at rp_plan_join)
    int n_marks = 0;                // host marks (rp_plan_host_mark): the plan replays as n_marks + 1 segments
    hipStream_t side2 = nullptr;
    hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr, ev_sync2 = nullptr;
    bool own_side = false, own_side2 = false;  // streams created here (else: the caller's, rp_plan_set_streams)
    // probe (rp_plan_set_probe): ONE launch of the next replays is bracketed by a timing event pair on the stream it is
    // issued on — the launch's duration INSIDE the replayed step, beside whatever the side streams run (bench.py)
    int probe = -1;
    hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr;
    bool probe_recorded = false;
    // input rebinding (rp_plan_bind_inputs): 8-byte words of the packed arguments that hold the address of one of the caller's
    // static input buffers -> (offset into blob, input index); rp_plan_set_inputs writes other addresses there
    std::vector<std::pair<size_t, int>> binds;
    // the slowest single HIP call (launch / event record / stream wait) of the replays since rp_plan_slowest_call reset it:
    // a launch call that blocks inside the runtime stalls the host's run-ahead (bench.py: host_stall)
    double slow_ms = 0.0;
    int slow_node = -1, slow_kind = -1;  // kind: 0 launch, 1 event record, 2 stream wait
};

// Fork/join events order streams of ONE device: an agent-scope release is all their consumers need.  The default event
// carries a system-scope fence (L2 write-back + invalidate at the record) that the next launch on the stream queues behind;
// RP_PLAN_EVENT_FENCE=system brings it back.
inline unsigned plan_event_flags() {
    static const bool sys = getenv("RP_PLAN_EVENT_FENCE") && strcmp(getenv("RP_PLAN_EVENT_FENCE"), "system") == 0;
    return sys ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence);
}

inline double plan_now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

std::atomic<Plan *> g_recording{nullptr};
std::mutex g_mu;

}  // namespace

bool rp_plan_recording() { return g_recording.load(std::memory_order_acquire) != nullptr; }

void rp_plan_record(const void *func, dim3 grid, dim3 block, unsigned shmem, hipStream_t stream, const char *blob,
                    size_t blob_bytes, const size_t *offsets, int nargs) {
    std::lock_guard<std::mutex> lock(g_mu);
    Plan *p = g_recording.load(std::memory_order_acquire);
    if (p == nullptr) return;
    PlanNode n;
    n.func = func;
    n.grid = grid;
    n.block = block;
    n.shmem = shmem;
    n.section = p->section;
    n.rec_stream = stream;
    n.blob_at = (p->blob.size() + 15) & ~(size_t)15;
    p->blob.resize(n.blob_at + (blob_bytes ? blob_bytes : 1));
    memcpy(p->blob.data() + n.blob_at, blob, blob_bytes);
    n.offs.assign(offsets, offsets + nargs);
    if (n.section == 1) p->n_side++;
    if (n.section == 2) p->n_inline++;
    p->nodes.push_back(std::move(n));
}

extern "C" int rp_plan_begin(void **plan_out) {
    RP_REQUIRE(plan_out, "plan_begin: null pointer");
    std::lock_guard<std::mutex> lock(g_mu);
    RP_REQUIRE(g_recording.load() == nullptr, "plan_begin: another plan is being recorded");
    Plan *p = new Plan();
    g_recording.store(p, std::memory_order_release);
    *plan_out = p;
    return RP_OK;
}

extern "C" int rp_plan_section(int section) {
    std::lock_guard<std::mutex> lock(g_mu);
    Plan *p = g_recording.load();
    RP_REQUIRE(p != nullptr, "plan_section: no plan is being recorded");
    RP_REQUIRE(section >= 0 && section <= 2, "plan_section: 0 (main), 1 (side, forked at the fork point) or 2 (inline fork)");
    p->section = section;
    return RP_OK;
}

// the launches recorded under section 2 since the last join run BESIDE the main launches recorded after them; the main
// stream waits for them here (a marker node: func == nullptr)
extern "C" int rp_plan_join(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    Plan *p = g_recording.load();
    RP_REQUIRE(p != nullptr, "plan_join: no plan is being recorded");
    PlanNode n;
    n.func = nullptr;
    n.grid = n.block = dim3(0, 0, 0);
    n.shmem = 0;
    n.section = -2;
    n.rec_stream = nullptr;
    n.blob_at = p->blob.size();
    p->nodes.push_back(std::move(n));
    return RP_OK;
}

// The inline section's fork point, explicitly: launches recorded under section 2 AFTER this mark (until the join) depend on
// what the main stream held HERE, not on the main launches recorded between the mark and them.  Round 5: the first layer's
// backward issues its long main-stream launch (rp_embed_grad_seg) FIRST and the short side launches behind it — issued the
// other way round, the side launches filled every CU and the main launch (77 KB of LDS per workgroup) started 58 us late
// (profiles/r05_trace_step.txt).  A marker node (func == nullptr, section -3).
extern "C" int rp_plan_fork2_mark(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    Plan *p = g_recording.load();
    RP_REQUIRE(p != nullptr, "plan_fork2_mark: no plan is being recorded");
    PlanNode n;
    n.func = nullptr;
    n.grid = n.block = dim3(0, 0, 0);
    n.shmem = 0;
    n.section = -3;
    n.rec_stream = nullptr;
    n.blob_at = p->blob.size();
    p->nodes.push_back(std::move(n));
    return RP_OK;
}

// The main stream waits HERE for the side section (1) of the replay it is part of — for a step that consumes the next
// batch's sort itself (the optimizer catch-up of the next batch's rows issued at the END of the step, graph_step.py
// "catch-up ahead").  Without this marker the side section is joined at the end of the replay.  Marker node, section -4.
extern "C" int rp_plan_join_side(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    Plan *p = g_recording.load();
    RP_REQUIRE(p != nullptr, "plan_join_side: no plan is being recorded");
    PlanNode n;
    n.func = nullptr;
    n.grid = n.block = dim3(0, 0, 0);
    n.shmem = 0;
    n.section = -4;
    n.rec_stream = nullptr;
    n.blob_at = p->blob.size();
    p->nodes.push_back(std::move(n));
    return RP_OK;
}

// The inline section (2) waits HERE for what the main stream holds at this point — a second dependency edge for a section
// that was forked earlier (round 6: the tiny tables' gradient is forked in front of the sample-major launch and runs beside
// it; the launches behind the sample-major one follow on the same side stream and need that launch).  While the section is
// not open yet this is the fork mark.  Marker node, section -5.
extern "C" int rp_plan_side2_sync(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    Plan *p = g_recording.load();
    RP_REQUIRE(p != nullptr, "plan_side2_sync: no plan is being recorded");
    PlanNode n;
    n.func = nullptr;
    n.grid = n.block = dim3(0, 0, 0);
    n.shmem = 0;
    n.section = -5;
    n.rec_stream = nullptr;
    n.blob_at = p->blob.size();
    p->nodes.push_back(std::move(n));
    return RP_OK;
}

// HOST MARKS (round 6: row-sharded steps).  A step whose launches are interleaved with work the library does not issue — the
// collectives of the row-sharded path: RCCL's all-to-alls and the dense all-reduce — is recorded as SEGMENTS: the caller marks
// every such point while recording (and does not issue the foreign work under the capture), replays segment k with
// rp_plan_replay_segment and issues the foreign work itself between two segments, on the same stream, with the same
// (capture-pool) buffers.  The device then sees the eager stream of kernels with the collectives where they were; the host
// pays a few microseconds per launch instead of the eager step's autograd / ctypes time.  Marker node, section -6;
// *index_out = the mark's number (0, 1, ...): segment k = the launches between mark k - 1 and mark k.
extern "C" int rp_plan_host_mark(int *index_out) {
    std::lock_guard<std::mutex> lock(g_mu);
    Plan *p = g_recording.load();
    RP_REQUIRE(p != nullptr, "plan_host_mark: no plan is being recorded");
    PlanNode n;
    n.func = nullptr;
    n.grid = n.block = dim3(0, 0, 0);
    n.shmem = 0;
    n.section = -6;
    n.rec_stream = nullptr;
    n.blob_at = p->blob.size();
    p->nodes.push_back(std::move(n));
    if (index_out != nullptr) *index_out = p->n_marks;
    p->n_marks++;
    return RP_OK;
}

extern "C" int rp_plan_host_marks(void *plan, int *n_marks) {
    Plan *p = reinterpret_cast<Plan *>(plan);
    RP_REQUIRE(p && n_marks, "plan_host_marks: null pointer");
    *n_marks = p->n_marks;
    return RP_OK;
}

extern "C" int rp_plan_is_recording(void) { return rp_plan_recording() ? 1 : 0; }

// the side section of a replay is forked HERE (in front of the next main-section launch) instead of at the start of the
// replay: the row sort of the next batch overlaps well with the backward's kernels and badly with the optimizer catch-up
// and the gather in front of them (measured on the eager path's side stream as well)
extern "C" int rp_plan_fork_here(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    Plan *p = g_recording.load();
    RP_REQUIRE(p != nullptr, "plan_fork_here: no plan is being recorded");
    p->fork_at = p->nodes.size();
    return RP_OK;
}

extern "C" int rp_plan_end(void *plan) {
    std::lock_guard<std::mutex> lock(g_mu);
    Plan *p = reinterpret_cast<Plan *>(plan);
    RP_REQUIRE(p != nullptr && g_recording.load() == p, "plan_end: this plan is not the one being recorded");
    g_recording.store(nullptr, std::memory_order_release);
    p->ptrs_at.clear();
    p->ptrs.clear();
    for (const PlanNode &n : p->nodes) {  // (blob no longer grows: the pointers are stable from here on)
        p->ptrs_at.push_back(p->ptrs.size());
        for (size_t o : n.offs) p->ptrs.push_back(p->blob.data() + n.blob_at + o);
        if (n.offs.empty()) p->ptrs.push_back(nullptr);
    }
    p->ended = true;
    return RP_OK;
}

// Input rebinding.  A recorded step reads its batch through pointers frozen in the launch arguments; graph_step.py used to keep
// two sets of static input buffers and copy every batch into one of them (one launch + 18 MB per step).  Instead: after
// rp_plan_end the caller names the addresses of those static buffers; every 8-byte word of the packed arguments that holds one
// of them is remembered, and rp_plan_set_inputs writes the addresses of the CURRENT batch's tensors there before a replay (the
// launch copies its arguments when it is issued).  Same shapes, dtypes and contiguity are the caller's business; so is keeping
// the tensors alive until the replay has run.  n_sites: how many words were found (0 = nothing to rebind).
extern "C" int rp_plan_bind_inputs(void *plan, const uint64_t *addrs, int n, int *n_sites) {
    Plan *p = reinterpret_cast<Plan *>(plan);
    RP_REQUIRE(p != nullptr && p->ended && addrs != nullptr && n >= 1, "plan_bind_inputs: a finished plan and >= 1 address");
    p->binds.clear();
    const size_t words = p->blob.size() / 8;
    for (size_t w = 0; w < words; ++w) {
        uint64_t v;
        memcpy(&v, p->blob.data() + 8 * w, 8);
        if (v == 0) continue;
        for (int i = 0; i < n; ++i)
            if (v == addrs[i]) {
                p->binds.emplace_back(8 * w, i);
                break;
            }
    }
    if (n_sites != nullptr) *n_sites = (int)p->binds.size();
    return RP_OK;
}

// What rp_plan_bind_inputs found, per input: sites[i] = argument words that hold EXACTLY addrs[i] (re-pointed by
// rp_plan_set_inputs), *n_interior = words that point INSIDE one of the buffers [addrs[i] + 1, addrs[i] + nbytes[i]) — a
// launch argument derived from an input (an offset view, a column of a packed buffer) that rp_plan_set_inputs can NOT
// re-point: the caller must keep the staging copy when there is one.
extern "C" int rp_plan_bind_report(void *plan, const uint64_t *addrs, const uint64_t *nbytes, int n, int32_t *sites,
                                   int *n_interior) {
    Plan *p = reinterpret_cast<Plan *>(plan);
    RP_REQUIRE(p != nullptr && p->ended && addrs != nullptr && nbytes != nullptr && n >= 1 && sites != nullptr && n_interior != nullptr,
               "plan_bind_report: a finished plan, >= 1 address, output pointers");
    for (int i = 0; i < n; ++i) sites[i] = 0;
    int interior = 0;
    const size_t words = p->blob.size() / 8;
    for (size_t w = 0; w < words; ++w) {
        uint64_t v;
        memcpy(&v, p->blob.data() + 8 * w, 8);
        if (v == 0) continue;
        for (int i = 0; i < n; ++i) {
            if (v == addrs[i]) {
                ++sites[i];
                break;
            }
            if (v > addrs[i] && v < addrs[i] + nbytes[i]) {
                ++interior;
                break;
            }
        }
    }
    *n_interior = interior;
    return RP_OK;
}

extern "C" int rp_plan_set_inputs(void *plan, const uint64_t *addrs, int n) {
    Plan *p = reinterpret_cast<Plan *>(plan);
    RP_REQUIRE(p != nullptr && p->ended && addrs != nullptr, "plan_set_inputs: null pointer");
    for (const auto &b : p->binds) {
        RP_REQUIRE(b.second < n, "plan_set_inputs: %d addresses, input %d is bound", n, b.second);
        memcpy(p->blob.data() + b.first, &addrs[b.second], 8);
    }
    return RP_OK;
}

// n_nodes: recorded launches; n_side: those of section 1; n_streams: distinct streams the launches were issued on while
// recording (graph_step.py records on ONE capture stream and refuses anything else)
extern "C" int rp_plan_info(void *plan, int *n_nodes, int *n_side, int *n_streams) {
    Plan *p = reinterpret_cast<Plan *>(plan);
    RP_REQUIRE(p && n_nodes && n_side && n_streams, "plan_info: null pointer");
    int launches = 0;
    for (const PlanNode &n : p->nodes) launches += n.func != nullptr ? 1 : 0;
    *n_nodes = launches;
    *n_side = p->n_side;
    std::vector<hipStream_t> seen;
    for (const PlanNode &n : p->nodes) {
        if (n.func == nullptr) continue;
        bool found = false;
        for (hipStream_t s : seen) found = found || s == n.rec_stream;
        if (!found) seen.push_back(n.rec_stream);
    }
    *n_streams = (int)seen.size();
    return RP_OK;
}

// The streams the side section (1) and the inline section (2) are re-issued on.  HIP multiplexes streams onto a handful
// of hardware queues (4 by default): a stream the plan created itself landed on the MAIN stream's hardware queue in a
// process that already owned three others, and its launches ran in line with the main ones.  The caller therefore hands
// in the streams its eager path overlaps on (they demonstrably sit on other queues); NULL = create one.
extern "C" int rp_plan_set_streams(void *plan, rp_stream_t side, rp_stream_t side2) {
    Plan *p = reinterpret_cast<Plan *>(plan);
    RP_REQUIRE(p != nullptr && p->ev_fork == nullptr && p->ev_fork2 == nullptr, "plan_set_streams: before the first replay");
    p->side = (hipStream_t)side;
    p->side2 = (hipStream_t)side2;
    return RP_OK;
}

// A non-blocking stream of the LOWEST priority the device offers (for the side streams of a step: the next batch's sort and
// the side work of the first layer's backward run BESIDE the main stream's launches and should yield to them — torch only
// hands out normal- and high-priority streams).  The caller owns it (never destroyed: streams of a process lifetime).
extern "C" int rp_stream_create_low(void **stream_out) {
    RP_REQUIRE(stream_out != nullptr, "stream_create_low: null pointer");
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStream_t s = nullptr;
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least);
    if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "stream_create_low: %s", hipGetErrorString(e));
    *stream_out = s;
    return RP_OK;
}

extern "C" int rp_plan_inline_count(void *plan, int *n_inline) {
    Plan *p = reinterpret_cast<Plan *>(plan);
    RP_REQUIRE(p && n_inline, "plan_inline_count: null pointer");
    *n_inline = p->n_inline;
    return RP_OK;
}

// launch index (recorded order, join markers not counted) -> node index; -1 when there is none
static int plan_launch_node(const Plan *p, int launch) {
    int k = 0;
    for (size_t i = 0; i < p->nodes.size(); ++i) {
        if (p->nodes[i].func == nullptr) continue;
        if (k == launch) return (int)i;
        ++k;
    }
    return -1;
}

// Time launch `launch` (0 .. n_nodes - 1 of rp_plan_info, recorded order) of the following replays: a HIP-event pair on the
// stream the launch is issued on.  -1 switches the probe off.  Two event records on one stream cost the step about as much
// as two small kernels; everything else of the replay is unchanged, so the launch shares the device with the same
// neighbours as in an unprobed replay.
extern "C" int rp_plan_set_probe(void *plan, int launch) {
    Plan *p = reinterpret_cast<Plan *>(plan);
    RP_REQUIRE(p != nullptr && p->ended, "plan_set_probe: the plan was not finished with rp_plan_end");
    RP_REQUIRE(launch < 0 || plan_launch_node(p, launch) >= 0, "plan_set_probe: launch %d does not exist", launch);
    if (launch >= 0 && p->ev_p0 == nullptr) {
        hipError_t e = hipEventCreate(&p->ev_p0);
        if (e == hipSuccess) e = hipEventCreate(&p->ev_p1);
        if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_set_probe: %s", hipGetErrorString(e));
    }
    p->probe = launch < 0 ? -1 : plan_launch_node(p, launch);
    p->probe_recorded = false;
    return RP_OK;
}

// milliseconds between the probe's two events of the LAST replay (waits for the second one)
extern "C" int rp_plan_probe_ms(void *plan, float *ms) {
    Plan *p = reinterpret_cast<Plan *>(plan);
    RP_REQUIRE(p != nullptr && ms != nullptr, "plan_probe_ms: null pointer");
    RP_REQUIRE(p->probe >= 0 && p->probe_recorded, "plan_probe_ms: no probed replay yet");
    hipError_t e = hipEventSynchronize(p->ev_p1);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, p->ev_p0, p->ev_p1);
    if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_probe_ms: %s", hipGetErrorString(e));
    return RP_OK;
}

// the (demangled) kernel name and the section (0 main, 1 side, 2 inline) of launch `launch`
extern "C" int rp_plan_launch_name(void *plan, int launch, char *buf, int buf_len, int *section) {
    Plan *p = reinterpret_cast<Plan *>(plan);
    RP_REQUIRE(p != nullptr && buf != nullptr && buf_len > 1, "plan_launch_name: bad argument");
    const int i = plan_launch_node(p, launch);
    RP_REQUIRE(i >= 0, "plan_launch_name: launch %d does not exist", launch);
    const char *nm = hipKernelNameRefByPtr(p->nodes[i].func, p->nodes[i].rec_stream);
    if (nm == nullptr) nm = "?";
    int status = -1;
    char *dem = abi::__cxa_demangle(nm, nullptr, nullptr, &status);  // (as rocprofv3 prints it)
    strncpy(buf, (status == 0 && dem != nullptr) ? dem : nm, (size_t)buf_len - 1);
    buf[buf_len - 1] = 0;
    if (dem != nullptr) free(dem);
    if (section) *section = p->nodes[i].section;
    return RP_OK;
}

// A completion marker for the host's run-ahead bound (graph_step.py): no timing, no system-scope fence — the host only waits
// for it, it never reads device memory on its strength.
extern "C" int rp_marker_create(void **marker) {
    RP_REQUIRE(marker != nullptr, "marker_create: null pointer");
    hipEvent_t ev = nullptr;
    hipError_t e = hipEventCreateWithFlags(&ev, plan_event_flags());
    if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "marker_create: %s", hipGetErrorString(e));
    *marker = ev;
    return RP_OK;
}

extern "C" int rp_marker_record(void *marker, rp_stream_t stream) {
    RP_REQUIRE(marker != nullptr, "marker_record: null marker");
    hipError_t e = hipEventRecord((hipEvent_t)marker, (hipStream_t)stream);
    if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "marker_record: %s", hipGetErrorString(e));
    return RP_OK;
}

extern "C" int rp_marker_wait(void *marker) {
    RP_REQUIRE(marker != nullptr, "marker_wait: null marker");
    hipError_t e = hipEventSynchronize((hipEvent_t)marker);
    if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "marker_wait: %s", hipGetErrorString(e));
    return RP_OK;
}

extern "C" int rp_marker_destroy(void *marker) {
    if (marker != nullptr) (void)hipEventDestroy((hipEvent_t)marker);
    return RP_OK;
}

// the slowest HIP call inside the replays since the last reset: kind (0 launch, 1 event record, 2 stream wait), the node
// it belongs to and its host time; reset != 0 starts a new window
extern "C" int rp_plan_slowest_call(void *plan, int *kind, int *node, double *ms, int reset) {
    Plan *p = reinterpret_cast<Plan *>(plan);
    RP_REQUIRE(p != nullptr, "plan_slowest_call: null plan");
    if (kind != nullptr) *kind = p->slow_kind;
    if (node != nullptr) *node = p->slow_node;
    if (ms != nullptr) *ms = p->slow_ms;
    if (reset) {
        p->slow_ms = 0.0;
        p->slow_node = p->slow_kind = -1;
    }
    return RP_OK;
}

// segment `seg` (0 .. n_marks) of a plan with host marks: its launches, in RECORDED order, all on `stream` — the launches of
// the side / inline sections too (the recording order is the issue order on the one capture stream: a valid serial order; the
// sections' fork / join events would have to straddle the caller's foreign work).
extern "C" int rp_plan_replay_segment(void *plan, int seg, rp_stream_t stream) {
    Plan *p = reinterpret_cast<Plan *>(plan);
    RP_REQUIRE(p != nullptr && p->ended, "plan_replay_segment: the plan was not finished with rp_plan_end");
    RP_REQUIRE(seg >= 0 && seg <= p->n_marks, "plan_replay_segment: segment %d of %d", seg, p->n_marks + 1);
    hipStream_t s = (hipStream_t)stream;
    int cur = 0;
    for (size_t i = 0; i < p->nodes.size(); ++i) {
        const PlanNode &n = p->nodes[i];
        if (n.func == nullptr) {
            if (n.section == -6) ++cur;
            if (cur > seg) break;
            continue;
        }
        if (cur != seg) continue;
        if ((int)i == p->probe) (void)hipEventRecord(p->ev_p0, s);
        const double t0 = plan_now_ms();
        const hipError_t e = hipLaunchKernel(n.func, n.grid, n.block, p->ptrs.data() + p->ptrs_at[i], n.shmem, s);
        const double d = plan_now_ms() - t0;
        if (d > p->slow_ms) {
            p->slow_ms = d;
            p->slow_node = (int)i;
            p->slow_kind = 0;
        }
        if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay_segment: launch %zu: %s", i, hipGetErrorString(e));
        if ((int)i == p->probe) {
            (void)hipEventRecord(p->ev_p1, s);
            p->probe_recorded = true;
        }
        rp_count_launch();
    }
    return RP_OK;
}

extern "C" int rp_plan_replay(void *plan, rp_stream_t stream) {
    Plan *p = reinterpret_cast<Plan *>(plan);
    RP_REQUIRE(p != nullptr && p->ended, "plan_replay: the plan was not finished with rp_plan_end");
    RP_REQUIRE(p->n_marks == 0, "plan_replay: this plan has %d host marks — replay it segment by segment (rp_plan_replay_segment)", p->n_marks);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    const bool fork = p->n_side > 0;
    if (fork && p->ev_fork == nullptr) {
        if (p->side == nullptr) {
            e = hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking);
            p->own_side = true;
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_fork, plan_event_flags());
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_join, plan_event_flags());
        if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay: side stream: %s", hipGetErrorString(e));
    }
    if (p->n_inline > 0 && p->ev_fork2 == nullptr) {
        if (p->side2 == nullptr) {
            e = hipStreamCreateWithFlags(&p->side2, hipStreamNonBlocking);
            p->own_side2 = true;
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_fork2, plan_event_flags());
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_join2, plan_event_flags());
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_sync2, plan_event_flags());
        if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay: second side stream: %s", hipGetErrorString(e));
    }
    bool forked = false, open2 = false, main_since_fork2 = false, side_joined = false, marked2 = false;
    auto timed = [p](int kind, size_t node, auto &&call) -> hipError_t {
        const double t0 = plan_now_ms();
        const hipError_t r = call();
        const double d = plan_now_ms() - t0;
        if (d > p->slow_ms) {
            p->slow_ms = d;
            p->slow_node = (int)node;
            p->slow_kind = kind;
        }
        return r;
    };
    for (size_t i = 0; i <= p->nodes.size(); ++i) {
        if (fork && !forked && (i >= p->fork_at || i == p->nodes.size())) {
            // the side section: it depends on nothing this replay computes, only on what was enqueued before the replay
            forked = true;
            e = timed(1, i, [&] { return hipEventRecord(p->ev_fork, s); });
            if (e == hipSuccess) e = timed(2, i, [&] { return hipStreamWaitEvent(p->side, p->ev_fork, 0); });
            if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay: fork: %s", hipGetErrorString(e));
            for (size_t j = 0; j < p->nodes.size(); ++j) {
                const PlanNode &n = p->nodes[j];
                if (n.section != 1) continue;
                if ((int)j == p->probe) (void)hipEventRecord(p->ev_p0, p->side);
                e = timed(0, j, [&] { return hipLaunchKernel(n.func, n.grid, n.block, p->ptrs.data() + p->ptrs_at[j], n.shmem, p->side); });
                if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay: side launch %zu: %s", j, hipGetErrorString(e));
                if ((int)j == p->probe) {
                    (void)hipEventRecord(p->ev_p1, p->side);
                    p->probe_recorded = true;
                }
            }
            e = timed(1, i, [&] { return hipEventRecord(p->ev_join, p->side); });
            if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay: join record: %s", hipGetErrorString(e));
        }
        if (i == p->nodes.size()) break;
        const PlanNode &n = p->nodes[i];
        if (n.section == 1) continue;
        if (n.func == nullptr && n.section == -3) {  // explicit fork point of the inline section
            if (p->n_inline > 0 && !open2) {
                e = timed(1, i, [&] { return hipEventRecord(p->ev_fork2, s); });
                if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay: inline fork mark: %s", hipGetErrorString(e));
                marked2 = true;
            }
            continue;
        }
        if (n.func == nullptr && n.section == -5) {  // the inline section needs the main stream's results from here on
            if (p->n_inline > 0) {
                if (open2) {
                    e = timed(1, i, [&] { return hipEventRecord(p->ev_sync2, s); });
                    if (e == hipSuccess) e = timed(2, i, [&] { return hipStreamWaitEvent(p->side2, p->ev_sync2, 0); });
                    main_since_fork2 = false;
                } else {
                    e = timed(1, i, [&] { return hipEventRecord(p->ev_fork2, s); });
                    marked2 = true;
                }
                if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay: inline sync: %s", hipGetErrorString(e));
            }
            continue;
        }
        if (n.func == nullptr && n.section == -4) {  // the main stream needs the side section's results from here on
            if (forked && !side_joined) {
                e = timed(2, i, [&] { return hipStreamWaitEvent(s, p->ev_join, 0); });
                if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay: join of the side section: %s", hipGetErrorString(e));
                side_joined = true;
            }
            continue;
        }
        if (n.func == nullptr) {  // join marker of the inline section
            marked2 = false;
            if (open2) {
                // the side section (issued at its fork point, long done by now) is joined through the same wait: every
                // event operation on the main stream is a packet its next launch queues behind
                // (round 5: only on request — with the side streams at the lowest priority the sort is no longer "long done":
                //  it ended 10 us AFTER the first layer's backward and held the optimizer up by 34 us, profiles/r05 trace; nothing
                //  in the step needs it, so it is joined at the end of the replay below)
                static const bool join_sort_early = getenv("RP_PLAN_JOIN_SORT") && strcmp(getenv("RP_PLAN_JOIN_SORT"), "early") == 0;
                if (join_sort_early && forked && !side_joined) {
                    e = timed(2, i, [&] { return hipStreamWaitEvent(p->side2, p->ev_join, 0); });
                    if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay: join of the side section: %s", hipGetErrorString(e));
                    side_joined = true;
                }
                e = timed(1, i, [&] { return hipEventRecord(p->ev_join2, p->side2); });
                if (e == hipSuccess) e = timed(2, i, [&] { return hipStreamWaitEvent(s, p->ev_join2, 0); });
                if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay: inline join: %s", hipGetErrorString(e));
                open2 = false;
            }
            continue;
        }
        hipStream_t target = s;
        if (n.section == 2) {
            // fork: what was enqueued on the main stream so far is what these launches depend on — also when main launches
            // were recorded between two groups of inline launches (the tiny-table gradient reads the transposed weight that
            // the main stream produces after the weight gradient was forked)
            if (marked2) {  // (the event was recorded at the mark: main launches recorded since are NOT waited for)
                if (!open2) {
                    e = timed(2, i, [&] { return hipStreamWaitEvent(p->side2, p->ev_fork2, 0); });
                    if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay: inline fork (marked): %s", hipGetErrorString(e));
                    open2 = true;
                }
            } else if (!open2 || main_since_fork2) {
                e = timed(1, i, [&] { return hipEventRecord(p->ev_fork2, s); });
                if (e == hipSuccess) e = timed(2, i, [&] { return hipStreamWaitEvent(p->side2, p->ev_fork2, 0); });
                if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay: inline fork: %s", hipGetErrorString(e));
                open2 = true;
                main_since_fork2 = false;
            }
            target = p->side2;
        } else if (open2) {
            main_since_fork2 = true;
        }
        if ((int)i == p->probe) (void)hipEventRecord(p->ev_p0, target);
        e = timed(0, i, [&] { return hipLaunchKernel(n.func, n.grid, n.block, p->ptrs.data() + p->ptrs_at[i], n.shmem, target); });
        if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay: launch %zu: %s", i, hipGetErrorString(e));
        if ((int)i == p->probe) {
            (void)hipEventRecord(p->ev_p1, target);
            p->probe_recorded = true;
        }
        rp_count_launch();
    }
    const size_t i = p->nodes.size();
    if (open2) {  // (a section that was never joined explicitly joins at the end)
        e = timed(1, i, [&] { return hipEventRecord(p->ev_join2, p->side2); });
        if (e == hipSuccess) e = timed(2, i, [&] { return hipStreamWaitEvent(s, p->ev_join2, 0); });
        if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay: inline join: %s", hipGetErrorString(e));
    }
    if (fork && !side_joined) {
        e = timed(2, i, [&] { return hipStreamWaitEvent(s, p->ev_join, 0); });
        if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "plan_replay: join: %s", hipGetErrorString(e));
    }
    return RP_OK;
}

extern "C" int rp_plan_destroy(void *plan) {
    Plan *p = reinterpret_cast<Plan *>(plan);
    if (p == nullptr) return RP_OK;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (g_recording.load() == p) g_recording.store(nullptr, std::memory_order_release);
    }
    if (p->ev_fork != nullptr) {
        (void)hipStreamSynchronize(p->side);
        (void)hipEventDestroy(p->ev_fork);
        (void)hipEventDestroy(p->ev_join);
        if (p->own_side) (void)hipStreamDestroy(p->side);
    }
    if (p->ev_p0 != nullptr) {
        (void)hipEventDestroy(p->ev_p0);
        (void)hipEventDestroy(p->ev_p1);
    }
    if (p->ev_fork2 != nullptr) {
        (void)hipStreamSynchronize(p->side2);
        (void)hipEventDestroy(p->ev_fork2);
        (void)hipEventDestroy(p->ev_join2);
        if (p->ev_sync2) (void)hipEventDestroy(p->ev_sync2);
        if (p->own_side2) (void)hipStreamDestroy(p->side2);
    }
    delete p;
    return RP_OK;
}

// ---- one launch that copies up to RP_MAX_COPY buffers (the next batch into a captured step's static input buffers: 40
// separate device-to-device copies of 0.25 - 0.5 MB cost the stream ~0.25 ms per step in dispatch gaps, one launch ~10 us)
#define RP_MAX_COPY 96
struct CopyList {
    void *dst[RP_MAX_COPY];
    const void *src[RP_MAX_COPY];
    uint64_t bytes[RP_MAX_COPY];
};

__global__ __launch_bounds__(256) void multi_copy_kernel(CopyList c) {
    const int t = blockIdx.y;
    const uint64_t nb = c.bytes[t];
    char *d = reinterpret_cast<char *>(c.dst[t]);
    const char *s = reinterpret_cast<const char *>(c.src[t]);
    const bool wide = ((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15u) == 0;
    const uint64_t n16 = wide ? nb / 16 : 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256)
        reinterpret_cast<uint4 *>(d)[i] = reinterpret_cast<const uint4 *>(s)[i];
    for (uint64_t i = n16 * 16 + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (uint64_t)gridDim.x * 256) d[i] = s[i];
}

extern "C" int rp_multi_copy(void *const *dst_ptrs, const void *const *src_ptrs, const uint64_t *bytes, int n,
                             rp_stream_t stream) {
    RP_REQUIRE(n >= 0 && (n == 0 || (dst_ptrs && src_ptrs && bytes)), "multi_copy: bad argument");
    hipStream_t s = (hipStream_t)stream;
    for (int i0 = 0; i0 < n; i0 += RP_MAX_COPY) {
        CopyList c;
        const int m = n - i0 < RP_MAX_COPY ? n - i0 : RP_MAX_COPY;
        uint64_t big = 0;
        for (int i = 0; i < RP_MAX_COPY; ++i) {
            c.dst[i] = i < m ? dst_ptrs[i0 + i] : nullptr;
            c.src[i] = i < m ? src_ptrs[i0 + i] : nullptr;
            c.bytes[i] = i < m ? bytes[i0 + i] : 0;
            RP_REQUIRE(i >= m || c.bytes[i] == 0 || (c.dst[i] && c.src[i]), "multi_copy: null buffer %d", i0 + i);
            if (c.bytes[i] > big) big = c.bytes[i];
        }
        if (big == 0) continue;
        uint64_t gx = rp_cdiv((int64_t)big, 256 * 16 * 4);  // ~4 wide elements per thread
        if (gx > 64) gx = 64;
        if (gx < 1) gx = 1;
        hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)gx, (unsigned)m), dim3(256), 0, s, c);
        RP_LAUNCH_CHECK("multi_copy");
    }
    return RP_OK;
}

// kernel nodes / other nodes (memset, memcpy, host, ...) of a captured hipGraph (torch.cuda.CUDAGraph.raw_cuda_graph())
extern "C" int rp_graph_node_counts(void *graph, int *n_kernel, int *n_other) {
    RP_REQUIRE(graph && n_kernel && n_other, "graph_node_counts: null pointer");
    hipGraph_t g = reinterpret_cast<hipGraph_t>(graph);
    size_t n = 0;
    hipError_t e = hipGraphGetNodes(g, nullptr, &n);
    if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "graph_node_counts: %s", hipGetErrorString(e));
    std::vector<hipGraphNode_t> nodes(n);
    if (n > 0) {
        e = hipGraphGetNodes(g, nodes.data(), &n);
        if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "graph_node_counts: %s", hipGetErrorString(e));
    }
    int k = 0, o = 0;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType t;
        e = hipGraphNodeGetType(nodes[i], &t);
        if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "graph_node_counts: %s", hipGetErrorString(e));
        if (t == hipGraphNodeTypeKernel) ++k;
        else if (t != hipGraphNodeTypeEmpty) ++o;
    }
    *n_kernel = k;
    *n_other = o;
    return RP_OK;
}
