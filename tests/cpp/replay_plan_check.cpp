// CPU check of fudanocr_amd/csrc/replay_plan.h (built by tests/test_host_logic.py with -fsanitize=address,undefined):
// on random DAGs -- chains with forks and joins like a captured training step, and dense random graphs -- every dependency
// of every node must be ORDERED before the node by the plan: same lane and earlier, or reachable through the waits
// (a wait on w orders its lane behind w, behind everything earlier on w's lane, and behind everything those were ordered
// behind).  Also: waits only name earlier nodes that record an event, lanes are within range, lane 0 is the longest chain.
#include "replay_plan.h"
#include <cstdio>
#include <cstdlib>
#include <random>

using focr_replay::Bits;

static bool check(const std::vector<std::vector<int>>& deps, int n_lanes, const char* what) {
  const focr_replay::Plan plan = focr_replay::plan_lanes(deps, n_lanes);
  const int n = (int)deps.size();
  // hb[i] = set of nodes complete before node i STARTS under the plan's stream semantics
  std::vector<Bits> hb(n, Bits(n));
  std::vector<int> last(n_lanes, -1);            // last node issued on each lane
  std::vector<int> cnt(n_lanes, 0);
  for (int i = 0; i < n; ++i) {
    const int L = plan.nodes[i].lane;
    if (L < 0 || L >= n_lanes || !plan.lane_used[L]) { std::printf("%s: node %d lane %d out of range\n", what, i, L); return false; }
    cnt[L]++;
    if (last[L] >= 0) { hb[i].orin(hb[last[L]]); hb[i].set(last[L]); }
    for (int w : plan.nodes[i].waits) {
      if (w >= i || !plan.nodes[w].record) { std::printf("%s: node %d waits for %d (not earlier / no event)\n", what, i, w); return false; }
      hb[i].orin(hb[w]);
      hb[i].set(w);
    }
    for (int d : deps[i])
      if (!hb[i].get(d)) { std::printf("%s: node %d (lane %d) is NOT ordered behind its dependency %d (lane %d)\n", what, i, L, d, plan.nodes[d].lane); return false; }
    last[L] = i;
    // a later wait on i also orders behind everything earlier on i's lane: already inside hb[i] (lane order above)
  }
  for (int L = 1; L < n_lanes; ++L)
    if (cnt[L] > cnt[L - 1]) { std::printf("%s: lanes not sorted by length\n", what); return false; }
  return true;
}

int main() {
  std::mt19937 rng(20261001);
  int graphs = 0;
  // (a) step-like graphs: a main chain, side work forked from main nodes, joins back, several side streams
  for (int rep = 0; rep < 300; ++rep) {
    const int n_streams = 1 + (int)(rng() % 5), n = 20 + (int)(rng() % 400), n_lanes = 1 + (int)(rng() % 6);
    std::vector<int> tail(n_streams, -1);
    std::vector<std::vector<int>> deps(n);
    for (int i = 0; i < n; ++i) {
      const int s = (rng() % 3 == 0) ? (int)(rng() % n_streams) : 0;
      if (tail[s] >= 0) deps[i].push_back(tail[s]);
      // cross-stream waits recorded during capture: depend on other streams' tails now and then
      for (int o = 0; o < n_streams; ++o)
        if (o != s && tail[o] >= 0 && rng() % 4 == 0 && std::find(deps[i].begin(), deps[i].end(), tail[o]) == deps[i].end())
          deps[i].push_back(tail[o]);
      tail[s] = i;
    }
    if (!check(deps, n_lanes, "step-like")) return 1;
    ++graphs;
  }
  // (b) dense random DAGs, including roots in the middle and more chains than lanes
  for (int rep = 0; rep < 300; ++rep) {
    const int n = 1 + (int)(rng() % 200), n_lanes = 1 + (int)(rng() % 4);
    std::vector<std::vector<int>> deps(n);
    for (int i = 1; i < n; ++i) {
      const int k = (int)(rng() % 4);
      for (int j = 0; j < k; ++j) {
        const int d = (int)(rng() % i);
        if (std::find(deps[i].begin(), deps[i].end(), d) == deps[i].end()) deps[i].push_back(d);
      }
    }
    if (!check(deps, n_lanes, "random")) return 1;
    ++graphs;
  }
  // (c) degenerate: no edges at all; one long chain
  {
    std::vector<std::vector<int>> none(17), chain(64);
    for (int i = 1; i < 64; ++i) chain[i].push_back(i - 1);
    if (!check(none, 3, "no edges") || !check(chain, 3, "chain")) return 1;
    const focr_replay::Plan p = focr_replay::plan_lanes(chain, 3);
    if (p.n_waits != 0) { std::printf("a chain needs no waits\n"); return 1; }
    graphs += 2;
  }
  std::printf("ok %d graphs\n", graphs);
  return 0;
}
