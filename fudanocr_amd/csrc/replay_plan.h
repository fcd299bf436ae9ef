// Lane plan of a recorded step (csrc/replay.hip): pure C++, no HIP -- the part of the replayer whose mistakes would be
// silent (a missing cross-lane wait is a data race, not an error code).  Kept in a header so that a CPU test can compile it
// with -fsanitize=address,undefined and check it on random DAGs (tests/cpp/replay_plan_check.cpp).
//
// Input: nodes 0 .. n-1 in a topological order (every dependency index is smaller than the node's), deps[i] = direct
// dependencies of node i, n_lanes = in-order queues available.  Output per node: its lane, the nodes whose completion event
// its lane must wait for before the node is issued, and whether an event has to be recorded behind it.
//
//   * a node goes (1) behind a direct dependency that is still the last node of its lane (lowest lane first), else (2) behind
//     the last node of any lane that is an ANCESTOR of it (lane order then adds no ordering the graph did not have), else
//     (3) on an unused lane, else (4) behind its latest dependency's lane (adds ordering, never removes any);
//   * a dependency on another lane needs a wait unless the node's lane is already ordered behind it: behind = it is in the
//     lane's ancestor set (everything the lane's earlier nodes, or the events they waited for, were ordered behind), or a
//     LATER node of the dependency's lane is (lanes are in order); per foreign lane only the latest dependency is waited for;
//   * finally lanes are renumbered by length: the longest chain becomes lane 0 (the caller's stream).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace focr_replay {

struct Bits {
  std::vector<uint64_t> w;
  explicit Bits(size_t n = 0) : w((n + 63) / 64, 0) {}
  void set(int i) { w[(size_t)i >> 6] |= 1ull << (i & 63); }
  bool get(int i) const { return (w[(size_t)i >> 6] >> (i & 63)) & 1; }
  void orin(const Bits& o) { for (size_t i = 0; i < w.size(); ++i) w[i] |= o.w[i]; }
};

struct PlanNode {
  int lane = 0;
  std::vector<int> waits;
  bool record = false;
};
struct Plan {
  std::vector<PlanNode> nodes;
  std::vector<int> lane_used;
  int n_waits = 0;
};

inline Plan plan_lanes(const std::vector<std::vector<int>>& deps, int n_lanes) {
  const size_t n = deps.size();
  Plan plan;
  plan.nodes.resize(n);
  plan.lane_used.assign(n_lanes, 0);
  std::vector<Bits> anc(n, Bits(n)), lane_anc(n_lanes, Bits(n));
  std::vector<int> tail(n_lanes, -1);
  for (size_t i = 0; i < n; ++i) {
    for (int d : deps[i]) { anc[i].orin(anc[d]); anc[i].set(d); }
    int lane = -1;
    for (int L = 0; L < n_lanes && lane < 0; ++L)
      if (tail[L] >= 0 && std::find(deps[i].begin(), deps[i].end(), tail[L]) != deps[i].end()) lane = L;
    for (int L = 0; L < n_lanes && lane < 0; ++L)
      if (tail[L] >= 0 && anc[i].get(tail[L])) lane = L;
    for (int L = 0; L < n_lanes && lane < 0; ++L)
      if (tail[L] < 0) lane = L;
    if (lane < 0) lane = deps[i].empty() ? 0 : plan.nodes[*std::max_element(deps[i].begin(), deps[i].end())].lane;
    PlanNode& nd = plan.nodes[i];
    nd.lane = lane;
    plan.lane_used[lane] = 1;
    std::vector<int> need(n_lanes, -1);
    for (int d : deps[i]) {
      const int dl = plan.nodes[d].lane;
      if (dl == lane || lane_anc[lane].get(d)) continue;
      need[dl] = std::max(need[dl], d);
    }
    for (int L = 0; L < n_lanes; ++L) {
      if (need[L] < 0) continue;
      bool covered = false;
      for (int t = tail[L]; t > need[L] && !covered; --t)
        if (plan.nodes[t].lane == L && lane_anc[lane].get(t)) covered = true;
      if (!covered) nd.waits.push_back(need[L]);
    }
    for (int w : nd.waits) {
      plan.nodes[w].record = true;
      lane_anc[lane].orin(anc[w]);
      lane_anc[lane].set(w);
      plan.n_waits++;
    }
    lane_anc[lane].orin(anc[i]);
    lane_anc[lane].set((int)i);
    tail[lane] = (int)i;
  }
  // longest chain -> lane 0, next -> lane 1, ...
  std::vector<int> cnt(n_lanes, 0), order(n_lanes), to(n_lanes);
  for (const PlanNode& nd : plan.nodes) cnt[nd.lane]++;
  for (int L = 0; L < n_lanes; ++L) order[L] = L;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cnt[a] > cnt[b]; });
  for (int L = 0; L < n_lanes; ++L) to[order[L]] = L;
  for (PlanNode& nd : plan.nodes) nd.lane = to[nd.lane];
  std::vector<int> used(n_lanes, 0);
  for (int L = 0; L < n_lanes; ++L) used[to[L]] = plan.lane_used[L];
  plan.lane_used.swap(used);
  return plan;
}

}  // namespace focr_replay
