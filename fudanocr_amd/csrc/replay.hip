// Step replay: the launch sequence of one whole optimisation step, recorded ONCE by HIP stream capture and re-issued from a
// C loop inside the library -- one host call per step instead of ~360 launches from Python / autograd / ATen.
//
// Why not hipGraphLaunch: measured on this runtime (tools/dev/graph_probe.py) an instantiated graph of
// the step costs as much host time per launch as enqueueing the step from Python and serialises the two queues.  What the
// step needs from a graph is only its CONTENT: which kernels, with which arguments, in which order, with which cross-stream
// edges.  focr_replay_build reads exactly that out of the captured (never instantiated) hipGraph_t through the public node
// getters and lays the nodes out on a few in-order "lanes" (HIP streams the caller hands over); focr_replay_launch walks the
// list: hipLaunchKernel / hipMemsetAsync / hipMemcpyAsync on the node's lane, hipStreamWaitEvent in front of a node for every
// dependency that lives on another lane and is not already ordered before it, hipEventRecord behind a node somebody waits
// for.  Lane 0 is the caller's stream: every lane starts behind it and is joined back into it at the end, so a replay is
// ordered against the caller's other work exactly like one kernel launch on that stream.
//
// Ownership (include/focr.h conventions): no device memory is allocated; the graph stays the caller's (it owns the kernel
// argument storage the node getters point into and must outlive the handle); the handle owns only HIP events.
#include "focr_common.h"
#include "replay_plan.h"
#include <vector>
#include <algorithm>
#include <cstring>
#include <cstdint>
typedef void* focr_stream_t;

namespace {

struct RNode {
  hipGraphNodeType type;
  hipKernelNodeParams kp;
  hipMemsetParams ms;
  hipMemcpy3DParms mc;
  int lane = 0;
  int module_launch = 0;          // kp.func is a hipFunction_t (captured hipModuleLaunchKernel), not a host stub
  std::vector<int> waits;         // node indices whose event this node's lane waits for first
  int event = -1;                 // index into Replay::events recorded behind this node
  int probe = -1;                 // index into Replay::probes: timing event pair around this node (focr_replay_probe)
};

struct Replay {
  std::vector<RNode> nodes;
  std::vector<hipStream_t> lanes;
  std::vector<hipEvent_t> events;
  std::vector<hipEvent_t> lane_end;
  std::vector<int> lane_used;
  hipEvent_t start = nullptr;
  struct Probe { std::vector<std::pair<hipEvent_t, hipEvent_t>> slots; long first_launch; };
  std::vector<Probe> probes;
  int counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long launches = 0;
};

#define RP_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      focr_set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e_));        \
      return FOCR_EHIP;                                                                   \
    }                                                                                     \
  } while (0)

void destroy(Replay* r) {
  if (!r) return;
  for (hipEvent_t e : r->events) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : r->lane_end) if (e) (void)hipEventDestroy(e);
  if (r->start) (void)hipEventDestroy(r->start);
  for (auto& p : r->probes)
    for (auto& e : p.slots) { if (e.first) (void)hipEventDestroy(e.first); if (e.second) (void)hipEventDestroy(e.second); }
  delete r;
}

}  // namespace

// graph: a hipGraph_t produced by stream capture (hipStreamEndCapture / torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()),
// NOT instantiated.  lanes[0] is the stream replays are ordered on; lanes[1..] carry the captured cross-stream concurrency.
// Kernel, memset, linear memcpy and empty nodes are understood; anything else (host callbacks, child graphs, external
// events, allocations) is refused with FOCR_EUNSUPPORTED and the caller keeps launching eagerly.
extern "C" int focr_replay_build(void* graph_, void* const* lanes, int n_lanes, void** handle) {
  FOCR_CHECK_ARG(graph_ && lanes && n_lanes >= 1 && handle, "null graph / lanes / handle");
  hipGraph_t graph = (hipGraph_t)graph_;
  size_t n = 0;
  RP_HIP(hipGraphGetNodes(graph, nullptr, &n));
  FOCR_CHECK_ARG(n > 0, "the captured graph is empty");
  std::vector<hipGraphNode_t> gn(n);
  RP_HIP(hipGraphGetNodes(graph, gn.data(), &n));
  gn.resize(n);
  // index of every node handle; dependencies must point BACKWARDS in the order we launch in.  The runtime hands the nodes
  // out in creation (= capture) order, which is a topological order; verified below and repaired by a stable topological
  // sort if a runtime ever does not.
  std::vector<std::vector<int>> deps(n);
  {
    std::vector<std::pair<hipGraphNode_t, int>> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = {gn[i], (int)i};
    std::sort(idx.begin(), idx.end());
    auto find = [&](hipGraphNode_t h) -> int {
      auto it = std::lower_bound(idx.begin(), idx.end(), std::make_pair(h, -1));
      return (it != idx.end() && it->first == h) ? it->second : -1;
    };
    bool ordered = true;
    for (size_t i = 0; i < n; ++i) {
      size_t nd = 0;
      RP_HIP(hipGraphNodeGetDependencies(gn[i], nullptr, &nd));
      if (!nd) continue;
      std::vector<hipGraphNode_t> d(nd);
      RP_HIP(hipGraphNodeGetDependencies(gn[i], d.data(), &nd));
      for (size_t k = 0; k < nd; ++k) {
        int j = find(d[k]);
        FOCR_CHECK_ARG(j >= 0, "a dependency is not a node of the graph");
        deps[i].push_back(j);
        if (j >= (int)i) ordered = false;
      }
    }
    if (!ordered) {                       // Kahn, smallest original index first
      std::vector<int> indeg(n, 0), order, pos(n);
      std::vector<std::vector<int>> succ(n);
      for (size_t i = 0; i < n; ++i) for (int j : deps[i]) { succ[j].push_back((int)i); ++indeg[i]; }
      std::vector<int> ready;
      for (size_t i = 0; i < n; ++i) if (!indeg[i]) ready.push_back((int)i);
      while (!ready.empty()) {
        auto it = std::min_element(ready.begin(), ready.end());
        int v = *it;
        ready.erase(it);
        order.push_back(v);
        for (int s : succ[v]) if (--indeg[s] == 0) ready.push_back(s);
      }
      FOCR_CHECK_ARG(order.size() == n, "the captured graph has a cycle");
      for (size_t i = 0; i < n; ++i) pos[order[i]] = (int)i;
      std::vector<hipGraphNode_t> gn2(n);
      std::vector<std::vector<int>> deps2(n);
      for (size_t i = 0; i < n; ++i) {
        gn2[i] = gn[order[i]];
        for (int j : deps[order[i]]) deps2[i].push_back(pos[j]);
      }
      gn.swap(gn2);
      deps.swap(deps2);
    }
  }
  Replay* r = new Replay();
  r->nodes.resize(n);
  r->lanes.assign((hipStream_t*)lanes, (hipStream_t*)lanes + n_lanes);
  r->lane_used.assign(n_lanes, 0);
  r->lane_end.assign(n_lanes, nullptr);
  for (size_t i = 0; i < n; ++i) {
    RNode& nd = r->nodes[i];
    hipError_t e = hipGraphNodeGetType(gn[i], &nd.type);
    if (e == hipSuccess) {
      switch (nd.type) {
        case hipGraphNodeTypeKernel:
          e = hipGraphKernelNodeGetParams(gn[i], &nd.kp);
          r->counts[1]++;
          break;
        case hipGraphNodeTypeMemset:
          e = hipGraphMemsetNodeGetParams(gn[i], &nd.ms);
          r->counts[2]++;
          break;
        case hipGraphNodeTypeMemcpy: {
          memset(&nd.mc, 0, sizeof(nd.mc));
          e = hipGraphMemcpyNodeGetParams(gn[i], &nd.mc);
          r->counts[3]++;
          // (a node captured from hipMemcpyAsync is a 1-D node whose parameters this getter does not fill on every
          // runtime: both pointers must be addresses the runtime knows, or the node is refused)
          hipPointerAttribute_t pa;
          if (e == hipSuccess && nd.mc.srcPtr.ptr && nd.mc.dstPtr.ptr && nd.mc.extent.width > 0 &&
              nd.mc.extent.width < (1ull << 40) &&
              (hipPointerGetAttributes(&pa, nd.mc.srcPtr.ptr) != hipSuccess ||
               hipPointerGetAttributes(&pa, nd.mc.dstPtr.ptr) != hipSuccess)) {
            (void)hipGetLastError();
            nd.mc.srcPtr.ptr = nullptr;
          }
          if (e == hipSuccess && (nd.mc.extent.width == 0 || nd.mc.extent.width >= (1ull << 40) || nd.mc.srcArray || nd.mc.dstArray || nd.mc.extent.height > 1 || nd.mc.extent.depth > 1 ||
                                  nd.mc.srcPos.x || nd.mc.srcPos.y || nd.mc.srcPos.z || nd.mc.dstPos.x || nd.mc.dstPos.y ||
                                  nd.mc.dstPos.z || !nd.mc.srcPtr.ptr || !nd.mc.dstPtr.ptr)) {
            focr_set_error("focr_replay_build: node %zu is a memcpy whose parameters cannot be read back (hipMemcpyAsync captures a 1-D node the "
                           "public getter does not describe) or is not a linear pointer-to-pointer copy: copy with a kernel inside a recorded step", i);
            destroy(r);
            return FOCR_EUNSUPPORTED;
          }
          break;
        }
        case hipGraphNodeTypeEmpty:
          r->counts[4]++;
          break;
        default:
          focr_set_error("focr_replay_build: node %zu has type %d (only kernel / memset / memcpy / empty nodes are replayed)", i,
                         (int)nd.type);
          destroy(r);
          return FOCR_EUNSUPPORTED;
      }
    }
    if (e != hipSuccess) {
      focr_set_error("focr_replay_build: reading node %zu (type %d) failed: %s", i, (int)nd.type, hipGetErrorString(e));
      destroy(r);
      return FOCR_EHIP;
    }
  }
  // ---- lanes and cross-lane waits: replay_plan.h (pure C++, also compiled and checked on the CPU under ASan / UBSan by
  // tests/test_host_logic.py::test_replay_plan_orders_every_dependency)
  {
    const focr_replay::Plan plan = focr_replay::plan_lanes(deps, n_lanes);
    for (size_t i = 0; i < n; ++i) {
      RNode& nd = r->nodes[i];
      nd.lane = plan.nodes[i].lane;
      nd.waits = plan.nodes[i].waits;
      if (plan.nodes[i].record) {
        nd.event = (int)r->events.size();
        r->events.push_back(nullptr);
      }
    }
    r->lane_used = plan.lane_used;
    r->counts[6] = plan.n_waits;
  }
  for (hipEvent_t& e : r->events) {
    hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (err != hipSuccess) {
      focr_set_error("focr_replay_build: hipEventCreate failed: %s", hipGetErrorString(err));
      destroy(r);
      return FOCR_EHIP;
    }
  }
  for (int L = 1; L < n_lanes; ++L)
    if (r->lane_used[L] && hipEventCreateWithFlags(&r->lane_end[L], hipEventDisableTiming) != hipSuccess) {
      focr_set_error("focr_replay_build: hipEventCreate failed");
      destroy(r);
      return FOCR_EHIP;
    }
  if (hipEventCreateWithFlags(&r->start, hipEventDisableTiming) != hipSuccess) {
    focr_set_error("focr_replay_build: hipEventCreate failed");
    destroy(r);
    return FOCR_EHIP;
  }
  r->counts[0] = (int)n;
  r->counts[5] = (int)std::count(r->lane_used.begin(), r->lane_used.end(), 1);
  r->counts[7] = (int)r->events.size();
  *handle = r;
  return FOCR_OK;
}

// out[8]: nodes, kernel nodes, memset nodes, memcpy nodes, empty nodes, lanes used, cross-lane waits, events recorded
extern "C" int focr_replay_info(void* handle, int* out) {
  FOCR_CHECK_ARG(handle && out, "null handle / out");
  memcpy(out, ((Replay*)handle)->counts, sizeof(int) * 8);
  return FOCR_OK;
}

// per-node table for tools (which lane every node landed on): lane[i] for i < min(n, nodes); returns FOCR_OK
extern "C" int focr_replay_lanes(void* handle, int* lane, int n) {
  FOCR_CHECK_ARG(handle && lane, "null handle / lane");
  Replay* r = (Replay*)handle;
  for (int i = 0; i < n && i < (int)r->nodes.size(); ++i) lane[i] = r->nodes[i].lane;
  return FOCR_OK;
}

// stream: the stream this replay is ordered on (it takes the place of lanes[0] of focr_replay_build for this launch); NULL =
// lanes[0] itself.
extern "C" int focr_replay_launch(void* handle, focr_stream_t stream) {
  FOCR_CHECK_ARG(handle, "null handle");
  Replay* r = (Replay*)handle;
  const int n_lanes = (int)r->lanes.size();
  hipStream_t lane0_saved = r->lanes[0];
  if (stream) r->lanes[0] = (hipStream_t)stream;
  struct Restore { Replay* r; hipStream_t s; ~Restore() { r->lanes[0] = s; } } restore{r, lane0_saved};
  RP_HIP(hipEventRecord(r->start, r->lanes[0]));
  for (int L = 1; L < n_lanes; ++L)
    if (r->lane_used[L]) RP_HIP(hipStreamWaitEvent(r->lanes[L], r->start, 0));
  const size_t n = r->nodes.size();
  for (size_t i = 0; i < n; ++i) {
    RNode& nd = r->nodes[i];
    hipStream_t s = r->lanes[nd.lane];
    for (int w : nd.waits) RP_HIP(hipStreamWaitEvent(s, r->events[r->nodes[w].event], 0));
    if (nd.probe >= 0) {
      Replay::Probe& p = r->probes[nd.probe];
      RP_HIP(hipEventRecord(p.slots[(r->launches - p.first_launch) % (long)p.slots.size()].first, s));
    }
    switch (nd.type) {
      case hipGraphNodeTypeKernel: {
        hipError_t e;
        if (!nd.module_launch) {
          e = hipLaunchKernel(nd.kp.func, nd.kp.gridDim, nd.kp.blockDim, nd.kp.kernelParams, nd.kp.sharedMemBytes, s);
          if (e == hipErrorInvalidDeviceFunction && r->launches == 0) {      // decided once, during the first replay
            (void)hipGetLastError();
            nd.module_launch = 1;
          } else if (e != hipSuccess) {
            focr_set_error("focr_replay_launch: node %zu: hipLaunchKernel failed: %s", i, hipGetErrorString(e));
            return FOCR_EHIP;
          }
        }
        if (nd.module_launch) {
          e = hipModuleLaunchKernel((hipFunction_t)nd.kp.func, nd.kp.gridDim.x, nd.kp.gridDim.y, nd.kp.gridDim.z,
                                    nd.kp.blockDim.x, nd.kp.blockDim.y, nd.kp.blockDim.z, nd.kp.sharedMemBytes, s,
                                    nd.kp.kernelParams, nd.kp.kernelParams ? nullptr : nd.kp.extra);
          if (e != hipSuccess) {
            focr_set_error("focr_replay_launch: node %zu: hipModuleLaunchKernel failed: %s", i, hipGetErrorString(e));
            return FOCR_EHIP;
          }
        }
        break;
      }
      case hipGraphNodeTypeMemset:
        if (nd.ms.height <= 1) {
          if (nd.ms.elementSize == 4) RP_HIP(hipMemsetD32Async((hipDeviceptr_t)nd.ms.dst, (int)nd.ms.value, nd.ms.width, s));
          else if (nd.ms.elementSize == 2) RP_HIP(hipMemsetD16Async((hipDeviceptr_t)nd.ms.dst, (unsigned short)nd.ms.value, nd.ms.width, s));
          else RP_HIP(hipMemsetD8Async((hipDeviceptr_t)nd.ms.dst, (unsigned char)nd.ms.value, nd.ms.width, s));
        } else {
          RP_HIP(hipMemset2DAsync(nd.ms.dst, nd.ms.pitch, (int)nd.ms.value, nd.ms.width * nd.ms.elementSize, nd.ms.height, s));
        }
        break;
      case hipGraphNodeTypeMemcpy:
        RP_HIP(hipMemcpyAsync(nd.mc.dstPtr.ptr, nd.mc.srcPtr.ptr, nd.mc.extent.width, nd.mc.kind, s));
        break;
      default:
        break;                   // empty node: only its edges matter
    }
    if (nd.probe >= 0) {
      Replay::Probe& p = r->probes[nd.probe];
      RP_HIP(hipEventRecord(p.slots[(r->launches - p.first_launch) % (long)p.slots.size()].second, s));
    }
    if (nd.event >= 0) RP_HIP(hipEventRecord(r->events[nd.event], s));
  }
  for (int L = 1; L < n_lanes; ++L)
    if (r->lane_used[L]) {
      RP_HIP(hipEventRecord(r->lane_end[L], r->lanes[L]));
      RP_HIP(hipStreamWaitEvent(r->lanes[0], r->lane_end[L], 0));
    }
  r->launches++;
  return FOCR_OK;
}

static const char* node_name(const RNode& nd) {
  if (nd.type == hipGraphNodeTypeMemset) return "(memset)";
  if (nd.type == hipGraphNodeTypeMemcpy) return "(memcpy)";
  if (nd.type != hipGraphNodeTypeKernel) return "(empty)";
  const char* nm = nd.module_launch ? hipKernelNameRef((hipFunction_t)nd.kp.func) : hipKernelNameRefByPtr(nd.kp.func, nullptr);
  return nm ? nm : "(unknown kernel)";
}

// (mangled) kernel name of node i, for tools and for choosing probe patterns; "(memset)" / "(memcpy)" / "(empty)" otherwise
extern "C" int focr_replay_node_name(void* handle, int i, char* buf, int n) {
  FOCR_CHECK_ARG(handle && buf && n > 0, "null handle / buf");
  Replay* r = (Replay*)handle;
  FOCR_CHECK_ARG(i >= 0 && i < (int)r->nodes.size(), "node index out of range");
  snprintf(buf, n, "%s", node_name(r->nodes[i]));
  return FOCR_OK;
}

// Timing probes: every kernel node whose name contains `pattern` gets `depth` timing-enabled event pairs; launch k records
// pair k % depth on the node's lane directly around its launch (bench.py's roofline leg: the dominant kernel's duration
// measured inside the timed region, on the stream it runs on).  Returns the number of nodes probed by this call (>= 0) or an
// error (< 0).
extern "C" int focr_replay_probe(void* handle, const char* pattern, int depth) {
  FOCR_CHECK_ARG(handle && pattern && *pattern && depth >= 1 && depth <= 4096, "null handle / pattern, or depth outside 1..4096");
  Replay* r = (Replay*)handle;
  int hit = 0;
  for (RNode& nd : r->nodes) {
    if (nd.type != hipGraphNodeTypeKernel || nd.probe >= 0 || !strstr(node_name(nd), pattern)) continue;
    Replay::Probe p;
    p.first_launch = r->launches;
    for (int k = 0; k < depth; ++k) {
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        focr_set_error("focr_replay_probe: hipEventCreate failed");
        return FOCR_EHIP;
      }
      p.slots.push_back({a, b});
    }
    nd.probe = (int)r->probes.size();
    r->probes.push_back(std::move(p));
    ++hit;
  }
  return hit;
}

// per probe k = 0 .. n-1 (node order over all focr_replay_probe calls): MEAN elapsed milliseconds over the launches its
// event pairs still hold (the last min(launches since the probe was set, depth)); the caller synchronises first.  node[k]
// (optional) receives the node index, count[k] (optional) the number of launches averaged.  Returns the number of probes.
extern "C" int focr_replay_probe_read(void* handle, float* ms, int* node, int* count, int n) {
  FOCR_CHECK_ARG(handle, "null handle");
  Replay* r = (Replay*)handle;
  for (size_t i = 0; i < r->nodes.size(); ++i) {
    int k = r->nodes[i].probe;
    if (k < 0 || k >= n) continue;
    Replay::Probe& p = r->probes[k];
    if (node) node[k] = (int)i;
    long have = r->launches - p.first_launch;
    if (have > (long)p.slots.size()) have = (long)p.slots.size();
    if (count) count[k] = (int)have;
    if (ms) {
      double sum = 0.0;
      for (long j = 0; j < have; ++j) {
        float t = 0.f;
        hipError_t e = hipEventElapsedTime(&t, p.slots[j].first, p.slots[j].second);
        if (e != hipSuccess) {
          focr_set_error("focr_replay_probe_read: probe %d: %s", k, hipGetErrorString(e));
          return FOCR_EHIP;
        }
        sum += t;
      }
      ms[k] = have ? (float)(sum / have) : 0.f;
    }
  }
  return (int)r->probes.size();
}

extern "C" int focr_replay_destroy(void* handle) {
  destroy((Replay*)handle);
  return FOCR_OK;
}
