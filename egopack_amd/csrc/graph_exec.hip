// Segmented replay of a captured training step (egk_graph_plan_*).
//
// Why: the HIP runtime replays a captured graph whose nodes sit on ONE stream through a fast path (AQL packets built once,
// handed to the queue as a batch: 0.6 us of host time and 1.7 us of device time per short launch on MI355X), but a graph with a
// single fork leaves that path -- every node is then enqueued by the host one by one, in creation order, chain after chain, at
// 2.8-4.2 us per node.  A step of 330 short launches on four streams (BASELINE config 4) needs 1.4 ms of host enqueueing for a
// 3.6 ms replay, and a branch that was created after another one starts when the host reaches it, not when its inputs are ready
// (tools/round4/chain_graph_probe.py: two independent chains of 200 launches as ONE graph run one after the other, 1269 us;
// as two single-stream graphs on two streams they overlap, 530 us).
//
// What: the captured graph (kernel / memset / memcpy / empty nodes) is cut into SEGMENTS -- maximal paths whose inner nodes have
// one predecessor and one successor -- every segment becomes its own single-stream hipGraph, and a replay launches the segments
// in creation order on up to max_streams streams, with an event wait for every edge that crosses streams and an event record
// behind every segment that has such a successor.  Same nodes, same edges (a cross-stream edge becomes record + wait; a
// same-stream edge becomes stream order), same kernel arguments: results are bit-identical to the runtime's replay.
//
// Stream choice follows the capture's own convention (the first-created successor of a fork keeps its predecessor's stream,
// cf. the fork-order rule of ops.defer_after_next_launch): a segment inherits the stream of a predecessor segment that is still
// the tail of its stream; otherwise it takes the stream whose tail is oldest.
#include <algorithm>
#include <map>
#include <queue>
#include <vector>

#include "common.h"

namespace {

struct Segment {
    std::vector<int> nodes;  // indices into the plan's node list, in chain order
    int stream = 0;          // 0 = the launch stream, i > 0 = side stream i - 1
    std::vector<int> wait;   // segments on OTHER streams this one waits for
    bool record = false;     // some successor segment runs on another stream
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipEvent_t done = nullptr;
};

}  // namespace

namespace {
// Event-node mode: consecutive segments of one stream form ONE graph whose cross-stream edges are event-record / event-wait
// NODES instead of stream operations between graph launches (a graph launch costs ~8 us of device time, a hand-off between two
// launches ~15).  The runtime binds an event-wait node to the event's most recent record when the node is ENQUEUED (at launch,
// in node order), so a piece must be launched after every piece that records an event it waits for.
struct Piece {
    int stream = 0;
    std::vector<int> segs;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};
}  // namespace

struct egk_graph_plan {
    std::vector<Piece> pieces;  // (event-node mode only)
    std::vector<Segment> segs;
    std::vector<hipStream_t> side;
    std::vector<hipEvent_t> side_done;
    hipEvent_t start = nullptr;
    int n_nodes = 0, n_edges = 0, n_cross = 0, n_streams = 1;
};

namespace {

#define EGK_HIP(call, what)                                                  \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) {                                              \
            ::egk::set_error("%s: %s", what, hipGetErrorString(e_));         \
            return (int)e_;                                                  \
        }                                                                    \
    } while (0)

int clone_node(hipGraph_t dst, hipGraphNode_t src, hipGraphNode_t prev, hipGraphNode_t* out) {
    hipGraphNodeType type;
    EGK_HIP(hipGraphNodeGetType(src, &type), "egk_graph_plan_create: hipGraphNodeGetType");
    const hipGraphNode_t* deps = prev ? &prev : nullptr;
    const size_t ndeps = prev ? 1 : 0;
    switch (type) {
        case hipGraphNodeTypeKernel: {
            hipKernelNodeParams p;
            EGK_HIP(hipGraphKernelNodeGetParams(src, &p), "egk_graph_plan_create: hipGraphKernelNodeGetParams");
            EGK_HIP(hipGraphAddKernelNode(out, dst, deps, ndeps, &p), "egk_graph_plan_create: hipGraphAddKernelNode");
            // launch attributes of the captured node (priority, cooperative launch, ...) travel with the clone; a runtime that
            // does not know the call leaves the defaults, which is what every launch of this library uses
            (void)hipGraphKernelNodeCopyAttributes(src, *out);
            return 0;
        }
        case hipGraphNodeTypeMemset: {
            hipMemsetParams p;
            EGK_HIP(hipGraphMemsetNodeGetParams(src, &p), "egk_graph_plan_create: hipGraphMemsetNodeGetParams");
            EGK_HIP(hipGraphAddMemsetNode(out, dst, deps, ndeps, &p), "egk_graph_plan_create: hipGraphAddMemsetNode");
            return 0;
        }
        case hipGraphNodeTypeMemcpy: {
            hipMemcpy3DParms p;
            EGK_HIP(hipGraphMemcpyNodeGetParams(src, &p), "egk_graph_plan_create: hipGraphMemcpyNodeGetParams");
            EGK_HIP(hipGraphAddMemcpyNode(out, dst, deps, ndeps, &p), "egk_graph_plan_create: hipGraphAddMemcpyNode");
            return 0;
        }
        case hipGraphNodeTypeEmpty:
            EGK_HIP(hipGraphAddEmptyNode(out, dst, deps, ndeps), "egk_graph_plan_create: hipGraphAddEmptyNode");
            return 0;
        default:
            ::egk::set_error("egk_graph_plan_create: node type %d is not supported (kernel, memset, memcpy and empty nodes are)", (int)type);
            return EGK_EINVAL;
    }
}

void destroy_plan(egk_graph_plan* p) {
    if (!p) return;
    for (auto& s : p->segs) {
        if (s.exec) (void)hipGraphExecDestroy(s.exec);
        if (s.graph) (void)hipGraphDestroy(s.graph);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    for (auto& pc : p->pieces) {
        if (pc.exec) (void)hipGraphExecDestroy(pc.exec);
        if (pc.graph) (void)hipGraphDestroy(pc.graph);
    }
    for (auto e : p->side_done) (void)hipEventDestroy(e);
    for (auto s : p->side) (void)hipStreamDestroy(s);
    if (p->start) (void)hipEventDestroy(p->start);
    delete p;
}

}  // namespace

extern "C" {

int egk_graph_plan_create(void* hip_graph, int32_t max_streams, int32_t event_nodes, egk_graph_plan** out) {
    EGK_REQUIRE(hip_graph && out && max_streams >= 1 && max_streams <= 16, "egk_graph_plan_create: bad arguments");
    *out = nullptr;
    hipGraph_t g = (hipGraph_t)hip_graph;
    size_t n = 0, ne = 0;
    EGK_HIP(hipGraphGetNodes(g, nullptr, &n), "egk_graph_plan_create: hipGraphGetNodes");
    EGK_REQUIRE(n > 0, "egk_graph_plan_create: the graph has no nodes");
    std::vector<hipGraphNode_t> nodes(n);
    EGK_HIP(hipGraphGetNodes(g, nodes.data(), &n), "egk_graph_plan_create: hipGraphGetNodes");
    EGK_HIP(hipGraphGetEdges(g, nullptr, nullptr, &ne), "egk_graph_plan_create: hipGraphGetEdges");
    std::vector<hipGraphNode_t> from(ne), to(ne);
    if (ne) EGK_HIP(hipGraphGetEdges(g, from.data(), to.data(), &ne), "egk_graph_plan_create: hipGraphGetEdges");
    std::map<hipGraphNode_t, int> index;
    for (size_t i = 0; i < n; ++i) index[nodes[i]] = (int)i;
    std::vector<std::vector<int>> pred(n), succ(n);
    for (size_t e = 0; e < ne; ++e) {
        auto a = index.find(from[e]), b = index.find(to[e]);
        EGK_REQUIRE(a != index.end() && b != index.end(), "egk_graph_plan_create: an edge names a node outside the graph");
        succ[a->second].push_back(b->second);
        pred[b->second].push_back(a->second);
    }
    for (size_t i = 0; i < n; ++i) {  // (duplicates would break the degree tests below)
        std::sort(pred[i].begin(), pred[i].end());
        pred[i].erase(std::unique(pred[i].begin(), pred[i].end()), pred[i].end());
        std::sort(succ[i].begin(), succ[i].end());
        succ[i].erase(std::unique(succ[i].begin(), succ[i].end()), succ[i].end());
    }
    // topological order, ties by node index (= creation order for a captured graph, which is itself topological)
    std::vector<int> order, indeg(n);
    {
        std::priority_queue<int, std::vector<int>, std::greater<int>> ready;
        for (size_t i = 0; i < n; ++i) {
            indeg[i] = (int)pred[i].size();
            if (!indeg[i]) ready.push((int)i);
        }
        while (!ready.empty()) {
            const int v = ready.top();
            ready.pop();
            order.push_back(v);
            for (int w : succ[v])
                if (--indeg[w] == 0) ready.push(w);
        }
        EGK_REQUIRE(order.size() == n, "egk_graph_plan_create: the graph has a cycle");
    }
    egk_graph_plan* plan = new egk_graph_plan();
    plan->n_nodes = (int)n;
    plan->n_edges = (int)ne;
    // segments: a node continues its predecessor's segment iff it is that node's only successor and has no other predecessor
    std::vector<int> seg_of(n, -1);
    for (int v : order) {
        if (pred[v].size() == 1 && succ[pred[v][0]].size() == 1) {
            seg_of[v] = seg_of[pred[v][0]];
        } else {
            seg_of[v] = (int)plan->segs.size();
            plan->segs.emplace_back();
        }
        plan->segs[seg_of[v]].nodes.push_back(v);
    }
    const int ns = (int)plan->segs.size();
    // streams: segments are visited in launch order (index order = order of their first nodes)
    std::vector<int> tail(max_streams, -1);      // last segment placed on each stream
    std::vector<int> placed_at(max_streams, -1);  // launch position of that segment (oldest tail = smallest)
    for (int s = 0; s < ns; ++s) {
        Segment& sg = plan->segs[s];
        std::vector<int> ps;
        for (int p : pred[sg.nodes.front()]) ps.push_back(seg_of[p]);
        std::sort(ps.begin(), ps.end());
        ps.erase(std::unique(ps.begin(), ps.end()), ps.end());
        int st = -1;
        for (int p : ps)
            if (tail[plan->segs[p].stream] == p) { st = plan->segs[p].stream; break; }
        if (st < 0) {
            for (int q = 0; q < max_streams; ++q)  // an unused stream, else the one whose tail is oldest
                if (st < 0 || placed_at[q] < placed_at[st]) st = q;
        }
        sg.stream = st;
        tail[st] = s;
        placed_at[st] = s;
        plan->n_streams = std::max(plan->n_streams, st + 1);
        for (int p : ps) {
            if (plan->segs[p].stream != st) {
                sg.wait.push_back(p);
                plan->segs[p].record = true;
                plan->n_cross++;
            }
        }
    }
    for (int s = 0; s < ns; ++s) {
        if (!plan->segs[s].record) continue;
        hipError_t e = hipEventCreateWithFlags(&plan->segs[s].done, hipEventDisableTiming);
        if (e != hipSuccess) { ::egk::set_error("egk_graph_plan_create: hipEventCreate: %s", hipGetErrorString(e)); destroy_plan(plan); return (int)e; }
    }
    if (event_nodes) {
        // pieces: a segment joins its stream's open piece unless one of the events it waits for is recorded by a piece that is
        // launched later than that piece
        std::vector<int> open(max_streams, -1), piece_of(ns, -1);
        for (int s = 0; s < ns; ++s) {
            Segment& sg = plan->segs[s];
            int pc = open[sg.stream];
            for (int w : sg.wait)
                if (pc >= 0 && piece_of[w] >= pc) pc = -1;
            if (pc < 0) {
                pc = (int)plan->pieces.size();
                plan->pieces.emplace_back();
                plan->pieces[pc].stream = sg.stream;
                open[sg.stream] = pc;
            }
            piece_of[s] = pc;
            plan->pieces[pc].segs.push_back(s);
        }
        for (Piece& pc : plan->pieces) {
            hipError_t e = hipGraphCreate(&pc.graph, 0);
            if (e != hipSuccess) { ::egk::set_error("egk_graph_plan_create: hipGraphCreate: %s", hipGetErrorString(e)); destroy_plan(plan); return (int)e; }
            hipGraphNode_t prev = nullptr;
            for (int s : pc.segs) {
                Segment& sg = plan->segs[s];
                for (int w : sg.wait) {
                    hipGraphNode_t nn = nullptr;
                    e = hipGraphAddEventWaitNode(&nn, pc.graph, prev ? &prev : nullptr, prev ? 1 : 0, plan->segs[w].done);
                    if (e != hipSuccess) { ::egk::set_error("egk_graph_plan_create: hipGraphAddEventWaitNode: %s", hipGetErrorString(e)); destroy_plan(plan); return (int)e; }
                    prev = nn;
                }
                for (int v : sg.nodes) {
                    hipGraphNode_t nn = nullptr;
                    const int rc = clone_node(pc.graph, nodes[v], prev, &nn);
                    if (rc) { destroy_plan(plan); return rc; }
                    prev = nn;
                }
                if (sg.record) {
                    hipGraphNode_t nn = nullptr;
                    e = hipGraphAddEventRecordNode(&nn, pc.graph, prev ? &prev : nullptr, prev ? 1 : 0, sg.done);
                    if (e != hipSuccess) { ::egk::set_error("egk_graph_plan_create: hipGraphAddEventRecordNode: %s", hipGetErrorString(e)); destroy_plan(plan); return (int)e; }
                    prev = nn;
                }
            }
            e = hipGraphInstantiate(&pc.exec, pc.graph, nullptr, nullptr, 0);
            if (e != hipSuccess) { ::egk::set_error("egk_graph_plan_create: hipGraphInstantiate: %s", hipGetErrorString(e)); destroy_plan(plan); return (int)e; }
        }
    }
    // one single-stream graph per segment
    for (int s = 0; s < ns && !event_nodes; ++s) {
        Segment& sg = plan->segs[s];
        hipError_t e = hipGraphCreate(&sg.graph, 0);
        if (e != hipSuccess) { ::egk::set_error("egk_graph_plan_create: hipGraphCreate: %s", hipGetErrorString(e)); destroy_plan(plan); return (int)e; }
        hipGraphNode_t prev = nullptr;
        for (int v : sg.nodes) {
            hipGraphNode_t nn = nullptr;
            const int rc = clone_node(sg.graph, nodes[v], prev, &nn);
            if (rc) { destroy_plan(plan); return rc; }
            prev = nn;
        }
        e = hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0);
        if (e != hipSuccess) { ::egk::set_error("egk_graph_plan_create: hipGraphInstantiate: %s", hipGetErrorString(e)); destroy_plan(plan); return (int)e; }
    }
    hipError_t e = hipEventCreateWithFlags(&plan->start, hipEventDisableTiming);
    for (int q = 1; q < plan->n_streams && e == hipSuccess; ++q) {
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr;
        e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e == hipSuccess) { plan->side.push_back(st); e = hipEventCreateWithFlags(&ev, hipEventDisableTiming); }
        if (e == hipSuccess) plan->side_done.push_back(ev);
    }
    if (e != hipSuccess) { ::egk::set_error("egk_graph_plan_create: streams / events: %s", hipGetErrorString(e)); destroy_plan(plan); return (int)e; }
    *out = plan;
    return 0;
}

int egk_graph_plan_info(const egk_graph_plan* plan, int32_t* nodes, int32_t* segments, int32_t* streams, int32_t* cross_edges) {
    EGK_REQUIRE(plan, "egk_graph_plan_info: null plan");
    if (nodes) *nodes = plan->n_nodes;
    if (segments) *segments = (int32_t)(plan->pieces.empty() ? plan->segs.size() : plan->pieces.size());
    if (streams) *streams = plan->n_streams;
    if (cross_edges) *cross_edges = plan->n_cross;
    return 0;
}

/* segment s: nodes[0] = number of nodes, nodes[1] = stream, nodes[2] = number of waits, nodes[3] = records an event */
int egk_graph_plan_segment(const egk_graph_plan* plan, int32_t s, int32_t* desc) {
    EGK_REQUIRE(plan && desc && s >= 0 && s < (int)plan->segs.size(), "egk_graph_plan_segment: bad arguments");
    const Segment& sg = plan->segs[s];
    desc[0] = (int32_t)sg.nodes.size();
    desc[1] = sg.stream;
    desc[2] = (int32_t)sg.wait.size();
    desc[3] = sg.record ? 1 : 0;
    return 0;
}

int egk_graph_plan_launch(egk_graph_plan* plan, egk_stream_t stream) {
    EGK_REQUIRE(plan, "egk_graph_plan_launch: null plan");
    hipStream_t main = (hipStream_t)stream;
    auto stream_of = [&](int q) { return q == 0 ? main : plan->side[q - 1]; };
    if (!plan->side.empty()) {  // the side streams start behind whatever the launch stream holds
        EGK_HIP(hipEventRecord(plan->start, main), "egk_graph_plan_launch: hipEventRecord");
        for (hipStream_t s : plan->side) EGK_HIP(hipStreamWaitEvent(s, plan->start, 0), "egk_graph_plan_launch: hipStreamWaitEvent");
    }
    for (Piece& pc : plan->pieces) EGK_HIP(hipGraphLaunch(pc.exec, stream_of(pc.stream)), "egk_graph_plan_launch: hipGraphLaunch");
    for (Segment& sg : plan->segs) {
        if (!plan->pieces.empty()) break;
        hipStream_t s = stream_of(sg.stream);
        for (int w : sg.wait) EGK_HIP(hipStreamWaitEvent(s, plan->segs[w].done, 0), "egk_graph_plan_launch: hipStreamWaitEvent");
        EGK_HIP(hipGraphLaunch(sg.exec, s), "egk_graph_plan_launch: hipGraphLaunch");
        if (sg.record) EGK_HIP(hipEventRecord(sg.done, s), "egk_graph_plan_launch: hipEventRecord");
    }
    for (size_t q = 0; q < plan->side.size(); ++q) {  // ... and the launch stream ends behind all of them
        EGK_HIP(hipEventRecord(plan->side_done[q], plan->side[q]), "egk_graph_plan_launch: hipEventRecord");
        EGK_HIP(hipStreamWaitEvent(main, plan->side_done[q], 0), "egk_graph_plan_launch: hipStreamWaitEvent");
    }
    return 0;
}

void egk_graph_plan_destroy(egk_graph_plan* plan) { destroy_plan(plan); }

}  // extern "C"
