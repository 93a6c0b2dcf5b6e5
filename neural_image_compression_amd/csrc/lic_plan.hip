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
// KernelNodeGetParams) into a list of operations with their predecessors.  A SCHEDULE puts the operations in an
// issue order, each on one of two streams, and keeps one event record + wait per cross-stream edge that stream
// order does not already imply.  lic_plan_replay issues a schedule with plain hipLaunchKernel / hipMemsetAsync /
// hipMemcpyAsync calls: ~3 us per operation of host time.
// Two schedules:
//   * capture order (lic_plan_create): topological, ties by creation order = the order the eager step launched in;
//     an operation continues the stream of a predecessor that is still that stream's tail, and one whose
//     predecessors' streams have all moved on starts the other stream -- the eager step's two branches again.
//   * tuned (lic_plan_tune): every operation is timed once (one-stream replay, an event between operations), then
//     list-scheduled by longest remaining path onto the stream where it finishes first, under a model of the GPU as
//     "kernels of >= 192 workgroups take turns, smaller ones run beside anything"; the tuned schedule is kept only
//     if it measures faster end to end than the capture-order one.
// Either way every edge of the capture is honoured (same-stream order or an event), so results are those of the
// eager step bit for bit.  Nothing here knows the model: any capture of kernel, memset, memcpy and empty nodes on
// one device works.
#include "lic_common.h"
#include <cstring>

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct PlanOp {
  hipGraphNodeType type;
  hipKernelNodeParams kp;  // (the argument storage belongs to the graph: the caller keeps the graph alive)
  hipMemsetParams ms;
  hipMemcpy3DParms mc;
  std::vector<int> preds, succs;  // indices into lic_plan::ops (capture order: preds < own index)
  double us = 0.0;                // measured duration (lic_plan_tune)
  bool wide = false;              // fills the chip: does not overlap with another wide operation
};

constexpr int NS = 3;  // streams a schedule may use: the caller's and up to two more

struct Slot {
  int op;
  int stream;
  int record = -1;         // event recorded right after this operation (a successor on the other stream waits for it)
  std::vector<int> waits;  // events the operation's stream waits for first
};

struct Schedule {
  std::vector<Slot> slots;  // issue order
  int n_events = 0;
  int on_side = 0;
};

// events for every cross-stream edge that earlier waits do not cover
void add_events(const std::vector<PlanOp>& ops, Schedule& sc) {
  const size_t n = sc.slots.size();
  std::vector<int> slot_of(n, -1), pos(n, -1);
  int len[NS] = {0, 0, 0};
  int waited[NS][NS];  // waited[s][o]: position on stream o that stream s has already waited for
  for (auto& row : waited)
    for (int& w : row) w = -1;
  sc.n_events = 0;
  sc.on_side = 0;
  for (size_t k = 0; k < n; ++k) {
    Slot& sl = sc.slots[k];
    sl.record = -1;
    sl.waits.clear();
    slot_of[sl.op] = (int)k;
  }
  for (size_t k = 0; k < n; ++k) {
    Slot& sl = sc.slots[k];
    const int s = sl.stream;
    for (int q : ops[sl.op].preds) {
      Slot& pq = sc.slots[slot_of[q]];
      if (pq.stream == s || pos[slot_of[q]] <= waited[s][pq.stream]) continue;
      if (pq.record < 0) pq.record = sc.n_events++;
      sl.waits.push_back(pq.record);
      waited[s][pq.stream] = pos[slot_of[q]];
    }
    pos[k] = len[s]++;
    if (s != 0) ++sc.on_side;
  }
}

}  // namespace

struct lic_plan {
  std::vector<PlanOp> ops;  // topological (capture) order
  Schedule sched;
  std::vector<hipEvent_t> events;
  hipEvent_t fork = nullptr, join[NS - 1] = {nullptr, nullptr};
  int64_t count[4] = {0, 0, 0, 0};  // kernels, memsets, memcpys, empty
  int tuned = 0;
};

static thread_local std::string g_plan_error;

static int plan_fail(lic_plan* p, int rc, const std::string& why) {
  g_plan_error = why;
  if (p) lic_plan_destroy(p);
  return rc;
}

static bool ensure_events(lic_plan* p, int n) {
  while ((int)p->events.size() < n) {
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return false;
    p->events.push_back(ev);
  }
  return true;
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
  std::vector<int> order, rank(n, -1);
  order.reserve(n);
  while (!ready.empty()) {
    const int u = ready.top();
    ready.pop();
    rank[u] = (int)order.size();
    order.push_back(u);
    for (int v : succs[u])
      if (--indeg[v] == 0) ready.push(v);
  }
  if (order.size() != n) return plan_fail(nullptr, LIC_ERR_INVALID, "the graph has a cycle");

  lic_plan* p = new lic_plan();
  p->ops.resize(n);
  for (size_t k = 0; k < n; ++k) {
    const int u = order[k];
    PlanOp& op = p->ops[k];
    for (int q : preds[u]) op.preds.push_back(rank[q]);
    for (int q : succs[u]) op.succs.push_back(rank[q]);
    if (hipGraphNodeGetType(nodes[u], &op.type) != hipSuccess) return plan_fail(p, LIC_ERR_INVALID, "hipGraphNodeGetType failed");
    switch (op.type) {
      case hipGraphNodeTypeKernel:
        if (hipGraphKernelNodeGetParams(nodes[u], &op.kp) != hipSuccess) return plan_fail(p, LIC_ERR_INVALID, "hipGraphKernelNodeGetParams failed");
        if (!op.kp.func || (!op.kp.kernelParams && !op.kp.extra)) return plan_fail(p, LIC_ERR_UNSUPPORTED, "a kernel node without a function or arguments");
        op.wide = (long)op.kp.gridDim.x * op.kp.gridDim.y * op.kp.gridDim.z >= 192;
        ++p->count[0];
        break;
      case hipGraphNodeTypeMemset:
        if (hipGraphMemsetNodeGetParams(nodes[u], &op.ms) != hipSuccess) return plan_fail(p, LIC_ERR_INVALID, "hipGraphMemsetNodeGetParams failed");
        if (op.ms.height > 1) return plan_fail(p, LIC_ERR_UNSUPPORTED, "a 2-D memset node");
        ++p->count[1];
        break;
      case hipGraphNodeTypeMemcpy:
        if (hipGraphMemcpyNodeGetParams(nodes[u], &op.mc) != hipSuccess) return plan_fail(p, LIC_ERR_UNSUPPORTED, "a memcpy node whose parameters cannot be read back (1-D copy node)");
        if (getenv("LIC_PLAN_DEBUG"))
          fprintf(stderr, "[lic_plan] memcpy node %zu: src %p pitch %zu dst %p pitch %zu extent %zu x %zu x %zu kind %d\n", k,
                  op.mc.srcPtr.ptr, op.mc.srcPtr.pitch, op.mc.dstPtr.ptr, op.mc.dstPtr.pitch, op.mc.extent.width,
                  op.mc.extent.height, op.mc.extent.depth, (int)op.mc.kind);
        // (this ROCm answers hipSuccess for the 1-D copy node a captured hipMemcpyAsync becomes, with an unfilled struct)
        if (!op.mc.srcPtr.ptr || !op.mc.dstPtr.ptr || !op.mc.extent.width || (unsigned)op.mc.kind > (unsigned)hipMemcpyDefault ||
            op.mc.extent.width > ((size_t)1 << 40) || op.mc.extent.height > ((size_t)1 << 24) || op.mc.extent.depth > ((size_t)1 << 24))
          return plan_fail(p, LIC_ERR_UNSUPPORTED, "a memcpy node whose parameters this ROCm does not hand back (a captured hipMemcpyAsync: "
                                                   "a same-dtype contiguous tensor.copy_ / clone); replace the copy by a kernel");
        ++p->count[2];
        break;
      case hipGraphNodeTypeEmpty:
        ++p->count[3];
        break;
      default:
        return plan_fail(p, LIC_ERR_UNSUPPORTED, "node type " + std::to_string((int)op.type) + " (only kernel, memset, memcpy and empty nodes are replayed)");
    }
  }
  // capture-order schedule.  The capture's streams are chains of the graph (consecutive launches of a stream depend on
  // each other); where a chain forks, the successor with the longest way to go continues the stream and the others
  // start (or resume) another one: a stream that was never used, else the one that has been quiet longest.
  p->sched.slots.resize(n);
  std::vector<int> stream(n, 0), depth(n, 1);
  for (size_t k = n; k-- > 0;)
    for (int v : p->ops[k].succs) depth[k] = std::max(depth[k], depth[v] + 1);
  int tail[NS] = {-1, -1, -1};
  for (size_t k = 0; k < n; ++k) {
    int s = -1;
    for (int q : p->ops[k].preds) {
      if (tail[stream[q]] != q) continue;
      bool continues = true;   // k continues q's stream unless a sibling has the longer way to go
      for (int v : p->ops[q].succs)
        if (v != (int)k && (depth[v] > depth[k] || (depth[v] == depth[k] && v < (int)k))) continues = false;
      if (continues && (s < 0 || stream[q] < s)) s = stream[q];
    }
    if (s < 0) {
      if (p->ops[k].preds.empty()) {
        s = 0;
      } else {
        int latest = -1;
        for (int q : p->ops[k].preds) latest = std::max(latest, q);
        const int avoid = stream[latest];
        int pick = -1;
        for (int c = 0; c < NS; ++c)
          if (c != avoid && (pick < 0 || tail[c] < tail[pick])) pick = c;   // (-1 = never used sorts first)
        // the batched reduction forked off beside a chain (functional.flush_point) is a filler: it goes to the LAST stream --
        // the callers give the first extra stream a high priority for the chain it carries, and a high-priority launch
        // of ~6000 blocks takes the chip from the other chain's small launches (a 4 MB cast measured 69 us beside it)
        static const bool filler_last = [] {
          const char* e = getenv("LIC_PLAN_REDUCE_LAST");
          return !(e && e[0] == '0');
        }();
        if (filler_last && avoid != NS - 1 && p->ops[k].type == hipGraphNodeTypeKernel) {
          const char* nm = hipKernelNameRefByPtr(p->ops[k].kp.func, nullptr);
          if (nm && strstr(nm, "reduce_batch_kernel")) pick = NS - 1;
        }
        s = pick;
      }
    }
    stream[k] = s;
    tail[s] = (int)k;
    p->sched.slots[k].op = (int)k;
    p->sched.slots[k].stream = s;
  }
  add_events(p->ops, p->sched);
  if (!ensure_events(p, p->sched.n_events) || hipEventCreateWithFlags(&p->fork, hipEventDisableTiming) != hipSuccess)
    return plan_fail(p, LIC_ERR_LAUNCH, "hipEventCreate failed");
  for (auto& j : p->join)
    if (hipEventCreateWithFlags(&j, hipEventDisableTiming) != hipSuccess) return plan_fail(p, LIC_ERR_LAUNCH, "hipEventCreate failed");
  *out = p;
  return LIC_OK;
}

// info[0..6] = operations, kernels, memsets, memcpys, operations on the second stream, cross-stream events,
// 1 if the schedule in use is the tuned one
LIC_EXPORT int lic_plan_info(const lic_plan* p, int64_t* info) {
  if (!p || !info) return LIC_ERR_INVALID;
  info[0] = (int64_t)p->ops.size();
  info[1] = p->count[0];
  info[2] = p->count[1];
  info[3] = p->count[2];
  info[4] = p->sched.on_side;
  info[5] = p->sched.n_events;
  info[6] = p->tuned;
  return LIC_OK;
}

#define PLAN_TRY(expr)                                                                                          \
  do {                                                                                                          \
    const hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) {                                                                                     \
      g_lic_last_hip_error = (int)e_;                                                                           \
      g_plan_error = std::string(#expr) + " at operation " + std::to_string(k_) + ": " + hipGetErrorString(e_); \
      return LIC_ERR_LAUNCH;                                                                                    \
    }                                                                                                           \
  } while (0)

static int issue_op(const PlanOp& op, hipStream_t s, long k_) {
  switch (op.type) {
    case hipGraphNodeTypeKernel:
      if (op.kp.kernelParams) {
        PLAN_TRY(hipLaunchKernel(op.kp.func, op.kp.gridDim, op.kp.blockDim, op.kp.kernelParams, op.kp.sharedMemBytes, s));
      } else {
        PLAN_TRY(hipModuleLaunchKernel((hipFunction_t)op.kp.func, op.kp.gridDim.x, op.kp.gridDim.y, op.kp.gridDim.z,
                                       op.kp.blockDim.x, op.kp.blockDim.y, op.kp.blockDim.z, op.kp.sharedMemBytes, s,
                                       nullptr, op.kp.extra));
      }
      break;
    case hipGraphNodeTypeMemset:
      if (op.ms.elementSize == 4) PLAN_TRY(hipMemsetD32Async((hipDeviceptr_t)op.ms.dst, (int)op.ms.value, op.ms.width, s));
      else if (op.ms.elementSize == 2) PLAN_TRY(hipMemsetD16Async((hipDeviceptr_t)op.ms.dst, (unsigned short)op.ms.value, op.ms.width, s));
      else PLAN_TRY(hipMemsetD8Async((hipDeviceptr_t)op.ms.dst, (unsigned char)op.ms.value, op.ms.width, s));
      break;
    case hipGraphNodeTypeMemcpy:
      // (a one-row extent without pitches is what a 1-D copy looks like; hipMemcpy3DAsync refuses it)
      if (!op.mc.srcArray && !op.mc.dstArray && op.mc.extent.height <= 1 && op.mc.extent.depth <= 1 &&
          !op.mc.srcPos.x && !op.mc.srcPos.y && !op.mc.srcPos.z && !op.mc.dstPos.x && !op.mc.dstPos.y && !op.mc.dstPos.z)
        PLAN_TRY(hipMemcpyAsync(op.mc.dstPtr.ptr, op.mc.srcPtr.ptr, op.mc.extent.width, hipMemcpyDefault, s));
      else
        PLAN_TRY(hipMemcpy3DAsync(&op.mc, s));
      break;
    default:
      break;
  }
  return LIC_OK;
}

static int replay_schedule(lic_plan* p, const Schedule& sc, hipStream_t main, const hipStream_t* sides, int n_sides) {
  // streams the caller did not provide fold onto the last one it did (no side stream at all: everything on `main`)
  hipStream_t st[NS];
  st[0] = main;
  for (int i = 1; i < NS; ++i) st[i] = (i <= n_sides && sides && sides[i - 1]) ? sides[i - 1] : st[i - 1];
  long k_ = -1;
  bool distinct[NS] = {false, false, false};   // side streams that are really other streams (each once)
  for (int i = 1; i < NS; ++i) {
    distinct[i] = st[i] != st[0];
    for (int j = 1; j < i; ++j)
      if (st[j] == st[i]) distinct[i] = false;
  }
  if (distinct[1] || distinct[2]) PLAN_TRY(hipEventRecord(p->fork, st[0]));
  for (int i = 1; i < NS; ++i)
    if (distinct[i]) PLAN_TRY(hipStreamWaitEvent(st[i], p->fork, 0));
  for (const Slot& sl : sc.slots) {
    ++k_;
    hipStream_t s = st[sl.stream];
    for (int w : sl.waits) PLAN_TRY(hipStreamWaitEvent(s, p->events[w], 0));   // (same stream after folding: a no-op for the GPU)
    const int rc = issue_op(p->ops[sl.op], s, sl.op);
    if (rc != LIC_OK) return rc;
    if (sl.record >= 0) PLAN_TRY(hipEventRecord(p->events[sl.record], s));
  }
  for (int i = 1; i < NS; ++i)
    if (distinct[i]) {
      PLAN_TRY(hipEventRecord(p->join[i - 1], st[i]));
      PLAN_TRY(hipStreamWaitEvent(st[0], p->join[i - 1], 0));
    }
  return LIC_OK;
}

// Issue the plan: `main` is the caller's stream (work queued on it before the call is ordered before the plan, work
// queued after the call is ordered after ALL of the plan); `sides`: up to two more streams of the same device that are
// otherwise idle (fewer than the schedule uses: the missing ones fold onto the last one given; none: one stream).
LIC_EXPORT int lic_plan_replay(lic_plan* p, lic_stream_t main, const lic_stream_t* sides, int32_t n_sides) {
  if (!p || n_sides < 0 || (n_sides > 0 && !sides)) return LIC_ERR_INVALID;
  return replay_schedule(p, p->sched, (hipStream_t)main, (const hipStream_t*)sides, n_sides);
}

// wall time of `reps` back-to-back replays of a schedule, in microseconds per replay (the device is idle before and after)
static int time_schedule(lic_plan* p, const Schedule& sc, hipStream_t main, const hipStream_t* sides, int n_sides, int reps, double* us) {
  long k_ = -1;
  hipEvent_t e0, e1;
  PLAN_TRY(hipEventCreate(&e0));
  PLAN_TRY(hipEventCreate(&e1));
  int rc = replay_schedule(p, sc, main, sides, n_sides);  // warm
  PLAN_TRY(hipStreamSynchronize(main));
  if (rc == LIC_OK) {
    PLAN_TRY(hipEventRecord(e0, main));
    for (int r = 0; r < reps && rc == LIC_OK; ++r) rc = replay_schedule(p, sc, main, sides, n_sides);
    PLAN_TRY(hipEventRecord(e1, main));
    PLAN_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    PLAN_TRY(hipEventElapsedTime(&ms, e0, e1));
    *us = (double)ms * 1e3 / reps;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

// Time every operation, list-schedule onto the two streams, keep the fastest of {capture order, tuned candidates}.
// Replays the plan a few dozen times (the step is re-computed from the same inputs: idempotent as long as the
// capture holds no in-place update of its own inputs -- the optimizer stays outside).
// result[0..3] = sum of operation times (us), capture-order step (us), tuned step (us), 1 if the tuned one was kept.
LIC_EXPORT int lic_plan_tune(lic_plan* p, lic_stream_t main_, const lic_stream_t* sides_, int32_t n_sides, double* result) {
  if (!p || !sides_ || n_sides < 1 || !sides_[0] || sides_[0] == main_) return LIC_ERR_INVALID;
  hipStream_t main = (hipStream_t)main_;
  const hipStream_t* sides = (const hipStream_t*)sides_;
  const int ns = std::min(NS, 1 + (int)n_sides);   // streams the tuned schedule may use
  const size_t n = p->ops.size();
  long k_ = -1;
  // ---- 1. operation times: one stream, an event between operations, best of three passes
  std::vector<hipEvent_t> ev(n + 1);
  for (auto& e : ev) PLAN_TRY(hipEventCreate(&e));
  for (PlanOp& op : p->ops) op.us = 1e30;
  int rc = LIC_OK;
  for (int pass = 0; pass < 3 && rc == LIC_OK; ++pass) {
    PLAN_TRY(hipEventRecord(ev[0], main));
    for (size_t k = 0; k < n && rc == LIC_OK; ++k) {
      rc = issue_op(p->ops[k], main, (long)k);
      PLAN_TRY(hipEventRecord(ev[k + 1], main));
    }
    PLAN_TRY(hipStreamSynchronize(main));
    for (size_t k = 0; k < n && rc == LIC_OK; ++k) {
      float ms = 0.f;
      PLAN_TRY(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
      p->ops[k].us = std::min(p->ops[k].us, std::max((double)ms * 1e3, 0.5));
    }
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (rc != LIC_OK) return rc;
  double total = 0.0;
  for (const PlanOp& op : p->ops) total += op.us;
  if (const char* dbg = getenv("LIC_PLAN_DEBUG"))
    if (dbg[0] == '2')   // the step as a list of stand-alone operation times (capture order)
      for (size_t k = 0; k < n; ++k) {
        const PlanOp& op = p->ops[k];
        const char* nm = op.type == hipGraphNodeTypeKernel ? hipKernelNameRefByPtr(op.kp.func, main) : "(memset / memcpy / empty)";
        fprintf(stderr, "[lic_plan] op %3zu %8.1f us  grid %6u  %s\n", k, op.us,
                op.type == hipGraphNodeTypeKernel ? op.kp.gridDim.x * op.kp.gridDim.y * op.kp.gridDim.z : 0u, nm ? nm : "?");
      }
  // ---- 2. list scheduling.  Priority = longest path to the end (own time included).  Repeatedly the (ready
  //         operation, stream) pair that can START earliest is placed (ties: the more critical operation; a
  //         cross-stream dependency costs SYNC_US, so a chain stays on its stream unless the other one is clearly
  //         free earlier) -- weight-gradient and reduction launches, which nothing but the optimizer waits for, fill
  //         whichever stream is idle instead of sitting in the data-gradient chain's queue.  Long kernels of >= 192
  //         workgroups take turns (two of them side by side gain nothing); candidates differ in what "long" means.
  const double SYNC_US = 6.0;
  std::vector<double> up(n, 0.0);
  for (size_t k = n; k-- > 0;) {
    double m = 0.0;
    for (int v : p->ops[k].succs) m = std::max(m, up[v]);
    up[k] = p->ops[k].us + m;
  }
  auto list_schedule = [&](double wide_min_us, double* modelled) {
    Schedule sc;
    sc.slots.reserve(n);
    std::vector<double> finish(n, 0.0);
    std::vector<int> stream(n, -1), missing(n, 0);
    std::vector<int> ready;
    for (size_t k = 0; k < n; ++k) {
      missing[k] = (int)p->ops[k].preds.size();
      if (!missing[k]) ready.push_back((int)k);
    }
    double avail[NS] = {0.0, 0.0, 0.0}, wide_until = 0.0;
    while (!ready.empty()) {
      // earliest possible start over all (ready operation, stream) pairs; among the pairs within 1 us of it the most
      // critical operation, then the earlier start, then stream 0
      auto start_of = [&](int k, int s2) {
        const PlanOp& op = p->ops[k];
        double start = avail[s2];
        for (int q : op.preds) start = std::max(start, finish[q] + (stream[q] != s2 ? SYNC_US : 0.0));
        if (op.wide && op.us >= wide_min_us) start = std::max(start, wide_until);
        return start;
      };
      double min_start = 1e300;
      for (int k : ready)
        for (int s2 = 0; s2 < ns; ++s2) min_start = std::min(min_start, start_of(k, s2));
      int bi = -1, bs = 0;
      double b_start = 1e300;
      for (size_t i = 0; i < ready.size(); ++i)
        for (int s2 = 0; s2 < ns; ++s2) {
          const double st = start_of(ready[i], s2);
          if (st > min_start + 1.0) continue;
          const bool better = bi < 0 || up[ready[i]] > up[ready[bi]] + 1e-9 ||
                              (ready[i] == ready[bi] && st < b_start - 1e-9);
          if (better) bi = (int)i, bs = s2, b_start = st;
        }
      const int k = ready[bi];
      const PlanOp& op = p->ops[k];
      double start = avail[bs];
      for (int q : op.preds) start = std::max(start, finish[q] + (stream[q] != bs ? SYNC_US : 0.0));
      if (op.wide && op.us >= wide_min_us) start = std::max(start, wide_until);
      stream[k] = bs;
      finish[k] = start + op.us;
      avail[bs] = finish[k];
      if (op.wide && op.us >= wide_min_us) wide_until = finish[k];
      Slot sl;
      sl.op = k;
      sl.stream = bs;
      sc.slots.push_back(sl);
      ready.erase(ready.begin() + bi);
      for (int v : op.succs)
        if (--missing[v] == 0) ready.push_back(v);
    }
    add_events(p->ops, sc);
    *modelled = std::max(avail[0], std::max(avail[1], avail[2]));
    return sc;
  };
  // ---- 3. measure: the capture-order schedule and the candidates, keep the fastest (a tuned one must win by 2 %)
  double t_cap = 0.0, t_tuned = 1e300, modelled_best = 0.0;
  if ((rc = time_schedule(p, p->sched, main, sides, n_sides, 6, &t_cap)) != LIC_OK) return rc;
  Schedule best_sc;
  const double wide_opts[3] = {25.0, 80.0, 1e30};
  for (double wm : wide_opts) {
    double modelled = 0.0, t = 0.0;
    Schedule cand = list_schedule(wm, &modelled);
    if (!ensure_events(p, std::max(cand.n_events, p->sched.n_events))) return LIC_ERR_LAUNCH;
    if ((rc = time_schedule(p, cand, main, sides, n_sides, 6, &t)) != LIC_OK) return rc;
    if (getenv("LIC_PLAN_DEBUG"))
      fprintf(stderr, "[lic_plan] candidate (long >= %.0f us): %.1f us measured, %.1f us modelled, %d on the second stream, %d events\n",
              wm, t, modelled, cand.on_side, cand.n_events);
    if (t < t_tuned) t_tuned = t, best_sc = cand, modelled_best = modelled;
  }
  const bool keep = t_tuned < 0.98 * t_cap;
  if (getenv("LIC_PLAN_DEBUG"))
    fprintf(stderr, "[lic_plan] operations %.1f us in sum; capture order %.1f us (%d on the second stream, %d events); best tuned %.1f us "
                    "(%d on the second stream, %d events, modelled %.1f us): %s\n", total, t_cap, p->sched.on_side, p->sched.n_events,
            t_tuned, best_sc.on_side, best_sc.n_events, modelled_best, keep ? "tuned kept" : "capture order kept");
  if (keep) {
    p->sched = best_sc;
    p->tuned = 1;
  }
  if (result) {
    result[0] = total;
    result[1] = t_cap;
    result[2] = t_tuned;
    result[3] = keep ? 1.0 : 0.0;
  }
  return LIC_OK;
}
#undef PLAN_TRY

LIC_EXPORT void lic_plan_destroy(lic_plan* p) {
  if (!p) return;
  for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
  if (p->fork) (void)hipEventDestroy(p->fork);
  for (hipEvent_t j : p->join)
    if (j) (void)hipEventDestroy(j);
  delete p;
}
