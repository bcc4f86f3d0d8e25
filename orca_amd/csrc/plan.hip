// Offline embedding-cache planner (host code): read_batches + MRD of the reference's utils/cache.py:19-166.
// Belady-style: after every batch keep, among {uncached out-of-batch neighbours, currently cached ids,
// batch targets}, the `budget` ids whose NEXT use as a sampled neighbour is closest.
//
// The reference walks dense [n_nodes] arrays every batch (np.where(cache_flag==1), np.zeros(n_nodes):
// O(num_batch * n_nodes)); here the cached set is kept as an explicit sorted list, so a batch costs
// O(candidates log candidates).  Ties in reuse distance: the reference's np.argsort is an unstable
// sort, so membership under ties that straddle the budget cut is implementation-defined upstream
// (SURVEY H11); this planner defines the order as (distance, node id) ascending.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include "common.h"

namespace {
struct Occ { int32_t batch; int32_t kind; };   // kind 0 = target, 1 = out-of-batch neighbour (cache.py:80-87)
constexpr int64_t MAX_DISTANCE = 100000000;    // cache.py:95
}

ORCA_API int orca_mrd_plan_host(int64_t E, int64_t bs, int K, const int64_t* src, const int64_t* dst,
                                const int32_t* nbr_src, const int32_t* nbr_dst, int64_t n_nodes, int64_t budget,
                                int64_t* plan_out, int64_t* plan_len, int64_t* stats, double* avg_reuse_distance) {
  ORCA_REQUIRE(E >= 0 && bs > 0 && K >= 1 && n_nodes > 0 && budget >= 0, "orca_mrd_plan_host: bad sizes");
  ORCA_REQUIRE(E == 0 || (src && dst && nbr_src && nbr_dst && plan_out && plan_len), "orca_mrd_plan_host: null pointer");
  const int64_t num_batch = (E + bs - 1) / bs;
  // ---- pass 1: read_batches (cache.py:19-89) ---------------------------------------------------
  std::vector<std::vector<int64_t>> target(num_batch), out_ngh(num_batch);
  std::vector<std::vector<Occ>> occur((size_t)n_nodes);
  std::vector<int32_t> mark((size_t)n_nodes, -1);    // mark[id] == b  <=>  id is a target of batch b
  for (int64_t b = 0; b < num_batch; ++b) {
    const int64_t s = b * bs, e = std::min(E, s + bs);
    std::vector<int64_t>& ut = target[b];
    for (int64_t i = s; i < e; ++i) { ut.push_back(src[i]); }
    for (int64_t i = s; i < e; ++i) { ut.push_back(dst[i]); }
    for (int64_t v : ut)
      if (v < 0 || v >= n_nodes) { orca_set_error("orca_mrd_plan_host: node id %lld out of range", (long long)v); return ORCA_ERANGE; }
    std::sort(ut.begin(), ut.end());
    ut.erase(std::unique(ut.begin(), ut.end()), ut.end());
    for (int64_t v : ut) mark[v] = (int32_t)b;
    std::vector<int64_t>& out = out_ngh[b];
    for (int side = 0; side < 2; ++side) {
      const int32_t* nb = side ? nbr_dst : nbr_src;
      for (int64_t i = s * K; i < e * K; ++i) {
        const int64_t v = nb[i];
        if (v == 0) continue;                                   // cache.py:51
        if (v < 0 || v >= n_nodes) { orca_set_error("orca_mrd_plan_host: neighbour id out of range"); return ORCA_ERANGE; }
        if (mark[v] != (int32_t)b) out.push_back(v);            // cache.py:58-67 (in-batch neighbours dropped)
      }
    }
    for (int64_t v : ut) occur[v].push_back(Occ{(int32_t)b, 0});
    std::vector<int64_t> uo(out);
    std::sort(uo.begin(), uo.end());
    uo.erase(std::unique(uo.begin(), uo.end()), uo.end());
    for (int64_t v : uo) occur[v].push_back(Occ{(int32_t)b, 1});
  }
  // ---- pass 2: MRD (cache.py:92-166) -----------------------------------------------------------
  std::vector<uint8_t> cache_flag((size_t)n_nodes, 0);
  std::vector<int64_t> time_flag((size_t)n_nodes, 0);
  std::vector<int32_t> pos((size_t)n_nodes, 0), fresh((size_t)n_nodes, -1);
  std::vector<int64_t> cached, cand, dist, order;
  int64_t n_reuse = 0, n_recompute = 0;
  double total_rd = 0.0;
  for (int64_t b = 0; b < num_batch; ++b) {
    cand.clear();
    for (int64_t v : out_ngh[b]) {
      if (cache_flag[v]) { ++n_reuse; total_rd += (double)(b - time_flag[v]); }
      else { ++n_recompute; cand.push_back(v); fresh[v] = (int32_t)b; }
    }
    for (int64_t v : target[b]) { cand.push_back(v); fresh[v] = (int32_t)b; }   // new_computed (:124-125)
    cand.insert(cand.end(), cached.begin(), cached.end());
    std::sort(cand.begin(), cand.end());
    cand.erase(std::unique(cand.begin(), cand.end()), cand.end());                // :126-127
    dist.resize(cand.size());
    for (size_t i = 0; i < cand.size(); ++i) {                                    // :129-146
      const std::vector<Occ>& oc = occur[cand[i]];
      int32_t p = pos[cand[i]];
      while ((size_t)p < oc.size() && oc[p].batch <= b) ++p;
      pos[cand[i]] = p;
      dist[i] = ((size_t)p == oc.size() || oc[p].kind == 0) ? MAX_DISTANCE + 1 : (int64_t)oc[p].batch - b;
    }
    order.resize(cand.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int64_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return dist[x] < dist[y]; });   // :150-152
    const int64_t n_keep = std::min<int64_t>(budget, (int64_t)cand.size());
    for (int64_t v : cached) cache_flag[v] = 0;                                   // :155-156
    cached.clear();
    for (int64_t i = 0; i < n_keep; ++i) {
      const int64_t v = cand[order[i]];
      plan_out[b * budget + i] = v;
      cache_flag[v] = 1;
      cached.push_back(v);
      if (fresh[v] == (int32_t)b) time_flag[v] = b;                               // :160-162
    }
    plan_len[b] = n_keep;
  }
  if (stats) { stats[0] = n_reuse; stats[1] = n_recompute; }
  if (avg_reuse_distance) *avg_reuse_distance = n_reuse ? total_rd / (double)n_reuse : (double)NAN;
  return ORCA_OK;
}
