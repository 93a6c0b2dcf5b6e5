// Launch plans: a captured training step replayed on TWO HIP streams without Python in the loop.
//
// The bf16 configurations are host-paced: a step is ~180 launches of 5-100 us, and the Python around each (autograd
// nodes, tensor allocation, ctypes) costs ~18 us -- 3.1 ms of a 3.8 ms step, against 0.7 ms inside this library and
// ~3 ms of GPU work.  HIP graphs remove the host cost, but this ROCm's executor walks a graph with more than one
// branch node by node from the host (measured: 0.09 ms enqueue / 4.47 ms for the single-stream capture of the
// step, 8 ms enqueue / 10 ms for the two-stream capture), and the step needs its two branches: the decoder beside
// the latent side, the weight gradients beside the data gradients.
// So the graph is used only as the RECORD of the step -- torch captures it (private memory pool: every pointer is
// the same on every replay), and lic_plan_create reads the nodes and edges back (hipGraphGetNodes / GetEdges /
// KernelNodeGetParams), orders them topologically in capture order, spreads them over two streams (a node follows
// the predecessor whose stream it continues; a node none of whose predecessors is a stream tail starts the other
// stream) and turns each remaining cross-stream edge into one event record + wait.  lic_plan_replay then issues
// plain hipLaunchKernel / hipMemsetAsync / hipMemcpy3DAsync calls: ~3 us per node of host time, the same
// two-stream overlap on the GPU as the eager step.
// Nothing here knows the model: any capture of kernel, memset, memcpy and empty nodes on one device works.
#include "lic_common.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct PlanNode {
  hipGraphNodeType type;
  int stream = 0;
  int record = -1;         // event recorded right after this node (a successor on the other stream waits for it)
  std::vector<int> waits;  // events the node's stream waits for before the node
  hipKernelNodeParams kp;  // (the argument storage belongs to the graph: the caller keeps the graph alive)
  hipMemsetParams ms;
  hipMemcpy3DParms mc;
};

}  // namespace

struct lic_plan {
  std::vector<PlanNode> nodes;  // in issue order
  std::vector<hipEvent_t> events;
  hipEvent_t fork = nullptr, join = nullptr;
  int64_t count[4] = {0, 0, 0, 0};  // kernels, memsets, memcpys, empty
  int64_t on_side = 0;
};

static thread_local std::string g_plan_error;

static int plan_fail(lic_plan* p, int rc, const std::string& why) {
  g_plan_error = why;
  if (p) lic_plan_destroy(p);
  return rc;
}

LIC_EXPORT const char* lic_plan_last_error(void) { return g_plan_error.c_str(); }

LIC_EXPORT int lic_plan_create(void* hip_graph, lic_plan** out) {
  if (!hip_graph || !out) return LIC_ERR_INVALID;
  *out = nullptr;
  hipGraph_t g = (hipGraph_t)hip_graph;
  size_t n = 0, ne = 0;
  if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess || n == 0) return plan_fail(nullptr, LIC_ERR_INVALID, "hipGraphGetNodes failed or the graph is empty");
  std::vector<hipGraphNode_t> nodes(n);
  if (hipGraphGetNodes(g, nodes.data(), &n) != hipSuccess) return plan_fail(nullptr, LIC_ERR_INVALID, "hipGraphGetNodes failed");
  if (hipGraphGetEdges(g, nullptr, nullptr, &ne) != hipSuccess) return plan_fail(nullptr, LIC_ERR_INVALID, "hipGraphGetEdges failed");
  std::vector<hipGraphNode_t> from(ne), to(ne);
  if (ne && hipGraphGetEdges(g, from.data(), to.data(), &ne) != hipSuccess) return plan_fail(nullptr, LIC_ERR_INVALID, "hipGraphGetEdges failed");
  std::unordered_map<hipGraphNode_t, int> index;
  for (size_t i = 0; i < n; ++i) index[nodes[i]] = (int)i;
  std::vector<std::vector<int>> preds(n), succs(n);
  std::vector<int> indeg(n, 0);
  for (size_t e = 0; e < ne; ++e) {
    auto a = index.find(from[e]), b = index.find(to[e]);
    if (a == index.end() || b == index.end()) return plan_fail(nullptr, LIC_ERR_INVALID, "an edge names a node that is not in the graph");
    preds[b->second].push_back(a->second);
    succs[a->second].push_back(b->second);
    ++indeg[b->second];
  }
  // topological order, ties broken by creation (= capture) order: the order the eager step issued its launches in
  std::priority_queue<int, std::vector<int>, std::greater<int>> ready;
  for (size_t i = 0; i < n; ++i)
    if (indeg[i] == 0) ready.push((int)i);
  std::vector<int> order;
  order.reserve(n);
  while (!ready.empty()) {
    const int u = ready.top();
    ready.pop();
    order.push_back(u);
    for (int v : succs[u])
      if (--indeg[v] == 0) ready.push(v);
  }
  if (order.size() != n) return plan_fail(nullptr, LIC_ERR_INVALID, "the graph has a cycle");

  lic_plan* p = new lic_plan();
  p->nodes.resize(n);
  std::vector<int> slot(n, -1), pos(n, -1);  // graph node -> plan slot; position of a plan slot in its stream's sequence
  int tail[2] = {-1, -1}, len[2] = {0, 0};
  int waited[2][2] = {{-1, -1}, {-1, -1}};   // waited[s][o]: position on stream o that stream s has already waited for
  for (size_t k = 0; k < n; ++k) {
    const int u = order[k];
    PlanNode& nd = p->nodes[k];
    slot[u] = (int)k;
    if (hipGraphNodeGetType(nodes[u], &nd.type) != hipSuccess) return plan_fail(p, LIC_ERR_INVALID, "hipGraphNodeGetType failed");
    switch (nd.type) {
      case hipGraphNodeTypeKernel:
        if (hipGraphKernelNodeGetParams(nodes[u], &nd.kp) != hipSuccess) return plan_fail(p, LIC_ERR_INVALID, "hipGraphKernelNodeGetParams failed");
        if (!nd.kp.func || (!nd.kp.kernelParams && !nd.kp.extra)) return plan_fail(p, LIC_ERR_UNSUPPORTED, "a kernel node without a function or arguments");
        ++p->count[0];
        break;
      case hipGraphNodeTypeMemset:
        if (hipGraphMemsetNodeGetParams(nodes[u], &nd.ms) != hipSuccess) return plan_fail(p, LIC_ERR_INVALID, "hipGraphMemsetNodeGetParams failed");
        if (nd.ms.height > 1) return plan_fail(p, LIC_ERR_UNSUPPORTED, "a 2-D memset node");
        ++p->count[1];
        break;
      case hipGraphNodeTypeMemcpy:
        if (hipGraphMemcpyNodeGetParams(nodes[u], &nd.mc) != hipSuccess) return plan_fail(p, LIC_ERR_UNSUPPORTED, "a memcpy node whose parameters cannot be read back (1-D copy node)");
        if (getenv("LIC_PLAN_DEBUG"))
          fprintf(stderr, "[lic_plan] memcpy node %zu: src %p pitch %zu dst %p pitch %zu extent %zu x %zu x %zu kind %d\n", k,
                  nd.mc.srcPtr.ptr, nd.mc.srcPtr.pitch, nd.mc.dstPtr.ptr, nd.mc.dstPtr.pitch, nd.mc.extent.width,
                  nd.mc.extent.height, nd.mc.extent.depth, (int)nd.mc.kind);
        // (this ROCm answers hipSuccess for the 1-D copy node a captured hipMemcpyAsync becomes, with an unfilled struct)
        if (!nd.mc.srcPtr.ptr || !nd.mc.dstPtr.ptr || !nd.mc.extent.width || (unsigned)nd.mc.kind > (unsigned)hipMemcpyDefault ||
            nd.mc.extent.width > ((size_t)1 << 40) || nd.mc.extent.height > ((size_t)1 << 24) || nd.mc.extent.depth > ((size_t)1 << 24))
          return plan_fail(p, LIC_ERR_UNSUPPORTED, "a memcpy node whose parameters this ROCm does not hand back (a captured hipMemcpyAsync: "
                                                   "a same-dtype contiguous tensor.copy_ / clone); replace the copy by a kernel");
        ++p->count[2];
        break;
      case hipGraphNodeTypeEmpty:
        ++p->count[3];
        break;
      default:
        return plan_fail(p, LIC_ERR_UNSUPPORTED, "node type " + std::to_string((int)nd.type) + " (only kernel, memset, memcpy and empty nodes are replayed)");
    }
    // stream: continue the chain of a predecessor that is still its stream's tail (stream 0 first when a join has both)
    int s = -1;
    for (int q : preds[u]) {
      const int sq = p->nodes[slot[q]].stream;
      if (tail[sq] == slot[q] && (s < 0 || sq < s)) s = sq;
    }
    if (s < 0) {
      if (preds[u].empty()) {
        s = 0;
      } else {  // a fork: every predecessor's stream has moved on -- start (or resume) the other stream
        int latest = -1;
        for (int q : preds[u]) latest = std::max(latest, slot[q]);
        s = 1 - p->nodes[latest].stream;
      }
    }
    nd.stream = s;
    for (int q : preds[u]) {
      PlanNode& pq = p->nodes[slot[q]];
      if (pq.stream == s || pos[slot[q]] <= waited[s][pq.stream]) continue;
      if (pq.record < 0) {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return plan_fail(p, LIC_ERR_LAUNCH, "hipEventCreate failed");
        pq.record = (int)p->events.size();
        p->events.push_back(ev);
      }
      nd.waits.push_back(pq.record);
      waited[s][pq.stream] = pos[slot[q]];
    }
    pos[k] = len[s]++;
    tail[s] = (int)k;
    if (s == 1) ++p->on_side;
  }
  if (hipEventCreateWithFlags(&p->fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&p->join, hipEventDisableTiming) != hipSuccess)
    return plan_fail(p, LIC_ERR_LAUNCH, "hipEventCreate failed");
  *out = p;
  return LIC_OK;
}

// info[0..5] = nodes, kernel nodes, memset nodes, memcpy nodes, nodes on the second stream, cross-stream events
LIC_EXPORT int lic_plan_info(const lic_plan* p, int64_t* info) {
  if (!p || !info) return LIC_ERR_INVALID;
  info[0] = (int64_t)p->nodes.size();
  info[1] = p->count[0];
  info[2] = p->count[1];
  info[3] = p->count[2];
  info[4] = p->on_side;
  info[5] = (int64_t)p->events.size();
  return LIC_OK;
}

// Issue the plan: `main` is the caller's stream (work queued on it before the call is ordered before the plan, work
// queued after the call is ordered after ALL of the plan), `side` a second stream of the same device that is otherwise
// idle.  side == main (or null) runs everything on one stream.
LIC_EXPORT int lic_plan_replay(lic_plan* p, lic_stream_t main, lic_stream_t side) {
  if (!p) return LIC_ERR_INVALID;
  hipStream_t st[2] = {(hipStream_t)main, (hipStream_t)(side ? side : main)};
  const bool two = st[0] != st[1];
  long k_ = -1;
#define PLAN_TRY(expr)                    \
  do {                                    \
    const hipError_t e_ = (expr);         \
    if (e_ != hipSuccess) {               \
      g_lic_last_hip_error = (int)e_;     \
      g_plan_error = std::string(#expr) + " at node " + std::to_string(k_) + ": " + hipGetErrorString(e_); \
      return LIC_ERR_LAUNCH;              \
    }                                     \
  } while (0)
  if (two) {
    PLAN_TRY(hipEventRecord(p->fork, st[0]));
    PLAN_TRY(hipStreamWaitEvent(st[1], p->fork, 0));
  }
  for (PlanNode& nd : p->nodes) {
    ++k_;
    hipStream_t s = st[nd.stream];
    if (two)
      for (int w : nd.waits) PLAN_TRY(hipStreamWaitEvent(s, p->events[w], 0));
    switch (nd.type) {
      case hipGraphNodeTypeKernel:
        if (nd.kp.kernelParams) {
          PLAN_TRY(hipLaunchKernel(nd.kp.func, nd.kp.gridDim, nd.kp.blockDim, nd.kp.kernelParams, nd.kp.sharedMemBytes, s));
        } else {
          PLAN_TRY(hipModuleLaunchKernel((hipFunction_t)nd.kp.func, nd.kp.gridDim.x, nd.kp.gridDim.y, nd.kp.gridDim.z,
                                         nd.kp.blockDim.x, nd.kp.blockDim.y, nd.kp.blockDim.z, nd.kp.sharedMemBytes, s,
                                         nullptr, nd.kp.extra));
        }
        break;
      case hipGraphNodeTypeMemset:
        if (nd.ms.elementSize == 4) PLAN_TRY(hipMemsetD32Async((hipDeviceptr_t)nd.ms.dst, (int)nd.ms.value, nd.ms.width, s));
        else if (nd.ms.elementSize == 2) PLAN_TRY(hipMemsetD16Async((hipDeviceptr_t)nd.ms.dst, (unsigned short)nd.ms.value, nd.ms.width, s));
        else PLAN_TRY(hipMemsetD8Async((hipDeviceptr_t)nd.ms.dst, (unsigned char)nd.ms.value, nd.ms.width, s));
        break;
      case hipGraphNodeTypeMemcpy:
        // (a captured hipMemcpyAsync comes back as a pitch-less one-row extent, which hipMemcpy3DAsync refuses)
        if (!nd.mc.srcArray && !nd.mc.dstArray && nd.mc.extent.height <= 1 && nd.mc.extent.depth <= 1 &&
            !nd.mc.srcPos.x && !nd.mc.srcPos.y && !nd.mc.srcPos.z && !nd.mc.dstPos.x && !nd.mc.dstPos.y && !nd.mc.dstPos.z)
          PLAN_TRY(hipMemcpyAsync(nd.mc.dstPtr.ptr, nd.mc.srcPtr.ptr, nd.mc.extent.width, hipMemcpyDefault, s));
        else
          PLAN_TRY(hipMemcpy3DAsync(&nd.mc, s));
        break;
      default:
        break;
    }
    if (two && nd.record >= 0) PLAN_TRY(hipEventRecord(p->events[nd.record], s));
  }
  if (two) {
    PLAN_TRY(hipEventRecord(p->join, st[1]));
    PLAN_TRY(hipStreamWaitEvent(st[0], p->join, 0));
  }
#undef PLAN_TRY
  return LIC_OK;
}

LIC_EXPORT void lic_plan_destroy(lic_plan* p) {
  if (!p) return;
  for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
  if (p->fork) (void)hipEventDestroy(p->fork);
  if (p->join) (void)hipEventDestroy(p->join);
  delete p;
}
