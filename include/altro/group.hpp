// include/altro/group.hpp -- altro::BatchGroup: several GPUs of one node behind the facade.
//
// The reference solves one problem on one thread (/root/reference/perf/benchmark_unicycle.cpp:18-43) and has no
// multi-device code; the batch of independent instances is this build's only parallel axis.  A BatchGroup shards it:
// one AugmentedLagrangianiLQR<n, m> per device (constructed by the caller with device_id = devices[part] on ITS block
// of the global batch: ShardRange), all solved at once (SolveAsync on every solver: one host thread per handle inside
// the library), and ONE RCCL all-gather over xGMI of the 32-byte per-instance records {cost, violation,
// iterations_total, status}.  Link libaltro_group.so (which links librccl.so) next to libaltro_hip.so.
#pragma once

#include <exception>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "altro/altro.hpp"
#include "altro_group.h"

namespace altro {

class BatchGroup {
 public:
  explicit BatchGroup(std::vector<int> devices) : devices_(std::move(devices)) {
    if (altro_group_create(devices_.data(), (int)devices_.size(), &g_) != ALTRO_OK)
      throw std::runtime_error(std::string("altro_group_create failed: ") + altro_group_last_error(nullptr));
  }
  ~BatchGroup() { altro_group_destroy(g_); }
  BatchGroup(const BatchGroup&) = delete;
  BatchGroup& operator=(const BatchGroup&) = delete;

  int NumDevices() const { return (int)devices_.size(); }
  int Device(int part) const { return devices_.at(part); }
  // contiguous block [lo, hi) of a global batch of `total` instances that part `part` of `parts` solves
  static std::pair<int, int> ShardRange(int total, int parts, int part) {
    int lo = 0, hi = 0;
    altro_group_shard_range(total, parts, part, &lo, &hi);
    return {lo, hi};
  }
  // the solver of part `part` (constructed with device_id = Device(part)); stays the caller's
  template <int n, int m>
  void Attach(int part, augmented_lagrangian::AugmentedLagrangianiLQR<n, m>& solver) {
    // (the library reads batch, device and dimensions from the handle itself and rejects a part whose (n, m, N) differ
    //  from the parts already attached: the gather buffers are sized from them)
    Check(altro_group_attach(g_, part, solver.Handle(), solver.BatchSize()), "altro_group_attach");
    n_ = n;
    m_ = m;
    knots_ = solver.NumSegments();
    if ((int)async_.size() <= part) async_.resize(part + 1);
    async_[part] = {[&solver] { solver.SolveAsync(); }, [&solver] { solver.Wait(); }};
  }
  // AugmentedLagrangianiLQR::Solve on every part at once, then the exchange
  void Solve() {
    // exception-safe: when part k fails to start, the parts already in flight are waited for before the error leaves
    // (a solve left pending would keep its handle busy for good)
    size_t started = 0;
    try {
      for (; started < async_.size(); ++started)
        if (async_[started].start) async_[started].start();
    } catch (...) {
      for (size_t i = 0; i < started; ++i) {
        try {
          async_[i].wait();
        } catch (...) {
        }
      }
      throw;
    }
    std::exception_ptr first;
    for (auto& a : async_) {
      try {
        if (a.wait) a.wait();
      } catch (...) {
        if (!first) first = std::current_exception();
      }
    }
    if (first) std::rethrow_exception(first);
    Check(altro_group_gather(g_), "altro_group_gather");
  }
  int TotalInstances() const { return altro_group_total(g_); }
  // [TotalInstances()][4] = {cost, violation, iterations_total, status} in part order, as device `part` received them
  std::vector<double> Results(int part = 0) {
    std::vector<double> out((size_t)TotalInstances() * 4);
    Check(altro_group_get_results(g_, part, out.data(), TotalInstances()), "altro_group_get_results");
    return out;
  }
  double GatherMilliseconds() const { return altro_group_gather_ms(g_); }
  // The optional second collective: every device receives the trajectories of all parts.
  // X[TotalInstances()][N+1][n], U[TotalInstances()][N][m] in part order, as device `part` received them.
  void GatherTrajectories() { Check(altro_group_gather_trajectories(g_, n_, m_, knots_), "altro_group_gather_trajectories"); }
  std::pair<std::vector<double>, std::vector<double>> Trajectories(int part = 0) {
    std::vector<double> X((size_t)TotalInstances() * (knots_ + 1) * n_), U((size_t)TotalInstances() * knots_ * m_);
    Check(altro_group_get_trajectories(g_, part, X.data(), U.data(), TotalInstances()), "altro_group_get_trajectories");
    return {std::move(X), std::move(U)};
  }
  double TrajectoryGatherMilliseconds() const { return altro_group_trajectory_gather_ms(g_); }
  altro_group Handle() { return g_; }

 private:
  struct Async {
    std::function<void()> start, wait;
  };
  void Check(altro_status st, const char* what) {
    if (st != ALTRO_OK) throw std::runtime_error(std::string(what) + " failed: " + altro_group_last_error(g_));
  }
  std::vector<int> devices_;
  altro_group g_ = nullptr;
  std::vector<Async> async_;
  int n_ = 0, m_ = 0, knots_ = 0;  // dimensions of the attached solvers (the last Attach)
};

}  // namespace altro
