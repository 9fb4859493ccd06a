// altro_group.cpp — libaltro_group.so: the multi-GPU result exchange for single-process C / C++ callers
// (include/altro_group.h).  One RCCL communicator per device of the node (ncclCommInitAll),
// one all-gather of 32-byte records per solve; the solves themselves run through the C-ABI of libaltro_hip.so.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "../../include/altro_group.h"

struct altro_group_s {
  std::vector<int> dev;
  std::vector<ncclComm_t> comm;
  std::vector<hipStream_t> stream;
  std::vector<altro_handle> handle;
  std::vector<altro_desc> desc;  // what each attached handle says about itself (altro_get_desc)
  std::vector<int> batch;
  std::vector<double*> d_send;  // [slot][4] records of the part, padded to the largest part
  std::vector<double*> d_recv;  // [ndev][slot][4]
  std::vector<double> part_ms;
  double gather_ms = 0.0, traj_gather_ms = 0.0;
  int slot = 0;  // records per part in the gather buffers (= largest batch)
  // trajectory gather (altro_group_gather_trajectories): [slot][N+1][n] / [slot][N][m] per part, padded like the records
  std::vector<double*> d_sendX, d_recvX, d_sendU, d_recvU;
  int tslot = 0, tn = 0, tm = 0, tN = 0;
  bool comms_ok = false;
  std::string err;
};

namespace {
std::string g_create_err;

#define GROUP_HIP(expr)                                                                                      \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess) {                                                                                  \
      g->err = std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " #expr " -> " + hipGetErrorString(e_); \
      return ALTRO_HIP_ERROR;                                                                                \
    }                                                                                                        \
  } while (0)
#define GROUP_NCCL(expr)                                                                                      \
  do {                                                                                                        \
    ncclResult_t r_ = (expr);                                                                                 \
    if (r_ != ncclSuccess) {                                                                                  \
      g->err = std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " #expr " -> " + ncclGetErrorString(r_); \
      return ALTRO_HIP_ERROR;                                                                                 \
    }                                                                                                         \
  } while (0)

void FreeBuffers(altro_group g) {
  for (size_t i = 0; i < g->dev.size(); ++i) {
    if (hipSetDevice(g->dev[i]) != hipSuccess) continue;
    if (g->d_send[i]) hipFree(g->d_send[i]);
    if (g->d_recv[i]) hipFree(g->d_recv[i]);
    g->d_send[i] = g->d_recv[i] = nullptr;
  }
  g->slot = 0;
}
void FreeTrajectoryBuffers(altro_group g) {
  for (size_t i = 0; i < g->dev.size(); ++i) {
    if (hipSetDevice(g->dev[i]) != hipSuccess) continue;
    for (std::vector<double*>* v : {&g->d_sendX, &g->d_recvX, &g->d_sendU, &g->d_recvU}) {
      if ((*v)[i]) hipFree((*v)[i]);
      (*v)[i] = nullptr;
    }
  }
  g->tslot = 0;
}

altro_status EnsureBuffers(altro_group g) {
  const int ndev = (int)g->dev.size();
  int slot = 0;
  for (int i = 0; i < ndev; ++i) {
    if (!g->handle[i]) {
      g->err = "part " + std::to_string(i) + " has no handle attached (altro_group_attach)";
      return ALTRO_NOT_READY;
    }
    slot = std::max(slot, g->batch[i]);
  }
  if (slot == g->slot) return ALTRO_OK;
  FreeBuffers(g);
  for (int i = 0; i < ndev; ++i) {
    GROUP_HIP(hipSetDevice(g->dev[i]));
    GROUP_HIP(hipMalloc((void**)&g->d_send[i], (size_t)slot * 4 * sizeof(double)));
    GROUP_HIP(hipMalloc((void**)&g->d_recv[i], (size_t)ndev * slot * 4 * sizeof(double)));
    GROUP_HIP(hipMemsetAsync(g->d_send[i], 0, (size_t)slot * 4 * sizeof(double), g->stream[i]));
    GROUP_HIP(hipStreamSynchronize(g->stream[i]));
  }
  g->slot = slot;
  return ALTRO_OK;
}
}  // namespace

extern "C" {

void altro_group_shard_range(int total, int parts, int part, int* lo, int* hi) {
  if (parts < 1) parts = 1;
  const int base = total / parts, rem = total % parts;
  const int l = part * base + std::min(part, rem);
  if (lo) *lo = l;
  if (hi) *hi = l + base + (part < rem ? 1 : 0);
}

altro_status altro_group_create(const int* device_ids, int ndev, altro_group* out) {
  if (!device_ids || ndev < 1 || !out) return ALTRO_INVALID_ARG;
  altro_group g = new altro_group_s();
  g->dev.assign(device_ids, device_ids + ndev);
  g->comm.assign(ndev, nullptr);
  g->stream.assign(ndev, nullptr);
  g->handle.assign(ndev, nullptr);
  g->desc.assign(ndev, altro_desc{});
  g->batch.assign(ndev, 0);
  g->d_send.assign(ndev, nullptr);
  g->d_recv.assign(ndev, nullptr);
  g->part_ms.assign(ndev, 0.0);
  g->d_sendX.assign(ndev, nullptr);
  g->d_recvX.assign(ndev, nullptr);
  g->d_sendU.assign(ndev, nullptr);
  g->d_recvU.assign(ndev, nullptr);
  auto fail = [&](const std::string& what) {
    g_create_err = what;
    altro_group_destroy(g);
    return ALTRO_HIP_ERROR;
  };
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have < 1) return fail("no HIP device (the product has no CPU path)");
  for (int i = 0; i < ndev; ++i) {
    if (device_ids[i] < 0 || device_ids[i] >= have) return fail("device id " + std::to_string(device_ids[i]) + " out of range");
    if (hipSetDevice(device_ids[i]) != hipSuccess || hipStreamCreateWithFlags(&g->stream[i], hipStreamNonBlocking) != hipSuccess)
      return fail("cannot create a stream on device " + std::to_string(device_ids[i]));
  }
  // (after the solver handles' own streams where the caller created those first: the collective's streams then queue up
  //  behind the solver's hardware queues and not the other way round -- DESIGN.md section 6)
  const ncclResult_t r = ncclCommInitAll(g->comm.data(), ndev, g->dev.data());
  if (r != ncclSuccess) return fail(std::string("ncclCommInitAll -> ") + ncclGetErrorString(r));
  g->comms_ok = true;
  *out = g;
  return ALTRO_OK;
}

int altro_group_size(altro_group g) { return g ? (int)g->dev.size() : 0; }
int altro_group_total(altro_group g) {
  int t = 0;
  if (g)
    for (int b : g->batch) t += b;
  return t;
}

// The gather buffers are sized from what the HANDLE says (altro_get_desc): altro_pack_results_device and
// altro_pack_trajectory_device write the handle's own batch x (N, n, m), so a caller-supplied size that disagrees
// would let them write past the group's buffers.
altro_status altro_group_attach(altro_group g, int part, altro_handle h, int batch) {
  if (!g || !h || part < 0 || part >= (int)g->dev.size() || batch < 1) return ALTRO_INVALID_ARG;
  altro_desc d{};
  if (altro_get_desc(h, &d) != ALTRO_OK) return ALTRO_INVALID_ARG;
  if (d.batch != batch) {
    g->err = "altro_group_attach: part " + std::to_string(part) + " was given batch " + std::to_string(batch) +
             " but its handle was created with batch " + std::to_string(d.batch);
    return ALTRO_INVALID_ARG;
  }
  if (d.device_id != g->dev[part]) {
    g->err = "altro_group_attach: the handle of part " + std::to_string(part) + " lives on device " + std::to_string(d.device_id) +
             ", the part on device " + std::to_string(g->dev[part]);
    return ALTRO_INVALID_ARG;
  }
  for (size_t i = 0; i < g->handle.size(); ++i) {
    if ((int)i == part || !g->handle[i]) continue;
    if (g->desc[i].n != d.n || g->desc[i].m != d.m || g->desc[i].N != d.N) {
      g->err = "altro_group_attach: part " + std::to_string(part) + " has (n, m, N) = (" + std::to_string(d.n) + ", " +
               std::to_string(d.m) + ", " + std::to_string(d.N) + "), part " + std::to_string(i) + " (" +
               std::to_string(g->desc[i].n) + ", " + std::to_string(g->desc[i].m) + ", " + std::to_string(g->desc[i].N) + ")";
      return ALTRO_INVALID_ARG;
    }
  }
  g->handle[part] = h;
  g->batch[part] = d.batch;
  g->desc[part] = d;
  return ALTRO_OK;
}

altro_status altro_group_gather(altro_group g) {
  if (!g) return ALTRO_INVALID_ARG;
  altro_status st = EnsureBuffers(g);
  if (st != ALTRO_OK) return st;
  const int ndev = (int)g->dev.size();
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < ndev; ++i) {
    // (returns after the pack kernel has run on the handle's own stream)
    st = altro_pack_results_device(g->handle[i], g->d_send[i]);
    if (st != ALTRO_OK) {
      g->err = std::string("altro_pack_results_device on part ") + std::to_string(i) + ": " + altro_last_error(g->handle[i]);
      return st;
    }
  }
  GROUP_NCCL(ncclGroupStart());
  for (int i = 0; i < ndev; ++i) {
    GROUP_HIP(hipSetDevice(g->dev[i]));
    GROUP_NCCL(ncclAllGather(g->d_send[i], g->d_recv[i], (size_t)g->slot * 4, ncclDouble, g->comm[i], g->stream[i]));
  }
  GROUP_NCCL(ncclGroupEnd());
  for (int i = 0; i < ndev; ++i) {
    GROUP_HIP(hipSetDevice(g->dev[i]));
    GROUP_HIP(hipStreamSynchronize(g->stream[i]));
  }
  g->gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return ALTRO_OK;
}

altro_status altro_group_gather_trajectories(altro_group g, int n, int m, int N) {
  if (!g || n < 1 || m < 1 || N < 1) return ALTRO_INVALID_ARG;
  const int ndev = (int)g->dev.size();
  int slot = 0;
  for (int i = 0; i < ndev; ++i) {
    if (!g->handle[i]) {
      g->err = "part " + std::to_string(i) + " has no handle attached (altro_group_attach)";
      return ALTRO_NOT_READY;
    }
    // (the buffers below are sized from n, m, N; the pack kernels write the handle's own dimensions)
    if (g->desc[i].n != n || g->desc[i].m != m || g->desc[i].N != N) {
      g->err = "altro_group_gather_trajectories: (n, m, N) = (" + std::to_string(n) + ", " + std::to_string(m) + ", " +
               std::to_string(N) + ") but the handle of part " + std::to_string(i) + " has (" + std::to_string(g->desc[i].n) +
               ", " + std::to_string(g->desc[i].m) + ", " + std::to_string(g->desc[i].N) + ")";
      return ALTRO_INVALID_ARG;
    }
    slot = std::max(slot, g->batch[i]);
  }
  const size_t xrow = (size_t)(N + 1) * n, urow = (size_t)N * m;
  if (slot != g->tslot || n != g->tn || m != g->tm || N != g->tN) {
    FreeTrajectoryBuffers(g);
    for (int i = 0; i < ndev; ++i) {
      GROUP_HIP(hipSetDevice(g->dev[i]));
      GROUP_HIP(hipMalloc((void**)&g->d_sendX[i], slot * xrow * sizeof(double)));
      GROUP_HIP(hipMalloc((void**)&g->d_recvX[i], (size_t)ndev * slot * xrow * sizeof(double)));
      GROUP_HIP(hipMalloc((void**)&g->d_sendU[i], slot * urow * sizeof(double)));
      GROUP_HIP(hipMalloc((void**)&g->d_recvU[i], (size_t)ndev * slot * urow * sizeof(double)));
      GROUP_HIP(hipMemsetAsync(g->d_sendX[i], 0, slot * xrow * sizeof(double), g->stream[i]));  // (the padding of a smaller part)
      GROUP_HIP(hipMemsetAsync(g->d_sendU[i], 0, slot * urow * sizeof(double), g->stream[i]));
      GROUP_HIP(hipStreamSynchronize(g->stream[i]));
    }
    g->tslot = slot;
    g->tn = n;
    g->tm = m;
    g->tN = N;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < ndev; ++i) {
    const altro_status st = altro_pack_trajectory_device(g->handle[i], g->d_sendX[i], g->d_sendU[i]);
    if (st != ALTRO_OK) {
      g->err = std::string("altro_pack_trajectory_device on part ") + std::to_string(i) + ": " + altro_last_error(g->handle[i]);
      return st;
    }
  }
  GROUP_NCCL(ncclGroupStart());
  for (int i = 0; i < ndev; ++i) {
    GROUP_HIP(hipSetDevice(g->dev[i]));
    GROUP_NCCL(ncclAllGather(g->d_sendX[i], g->d_recvX[i], slot * xrow, ncclDouble, g->comm[i], g->stream[i]));
    GROUP_NCCL(ncclAllGather(g->d_sendU[i], g->d_recvU[i], slot * urow, ncclDouble, g->comm[i], g->stream[i]));
  }
  GROUP_NCCL(ncclGroupEnd());
  for (int i = 0; i < ndev; ++i) {
    GROUP_HIP(hipSetDevice(g->dev[i]));
    GROUP_HIP(hipStreamSynchronize(g->stream[i]));
  }
  g->traj_gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return ALTRO_OK;
}

altro_status altro_group_get_trajectories(altro_group g, int part, double* X, double* U, int capacity_instances) {
  if (!g || (!X && !U) || part < 0 || part >= (int)g->dev.size()) return ALTRO_INVALID_ARG;
  const int ndev = (int)g->dev.size();
  if (capacity_instances < altro_group_total(g)) {
    g->err = "altro_group_get_trajectories: the buffers hold fewer instances than the global batch";
    return ALTRO_INVALID_ARG;
  }
  if (!g->tslot || !g->d_recvX[part]) {
    g->err = "no trajectories gathered yet (altro_group_gather_trajectories)";
    return ALTRO_NOT_READY;
  }
  const size_t xrow = (size_t)(g->tN + 1) * g->tn, urow = (size_t)g->tN * g->tm;
  GROUP_HIP(hipSetDevice(g->dev[part]));
  size_t o = 0;
  for (int i = 0; i < ndev; ++i) {  // one copy per part: the padding of the smaller parts is skipped
    if (X)
      GROUP_HIP(hipMemcpyAsync(X + o * xrow, g->d_recvX[part] + (size_t)i * g->tslot * xrow, g->batch[i] * xrow * sizeof(double),
                               hipMemcpyDeviceToHost, g->stream[part]));
    if (U)
      GROUP_HIP(hipMemcpyAsync(U + o * urow, g->d_recvU[part] + (size_t)i * g->tslot * urow, g->batch[i] * urow * sizeof(double),
                               hipMemcpyDeviceToHost, g->stream[part]));
    o += g->batch[i];
  }
  GROUP_HIP(hipStreamSynchronize(g->stream[part]));
  return ALTRO_OK;
}
double altro_group_trajectory_gather_ms(altro_group g) { return g ? g->traj_gather_ms : 0.0; }

altro_status altro_group_solve_al(altro_group g) {
  if (!g) return ALTRO_INVALID_ARG;
  const int ndev = (int)g->dev.size();
  for (int i = 0; i < ndev; ++i)
    if (!g->handle[i]) {
      g->err = "part " + std::to_string(i) + " has no handle attached (altro_group_attach)";
      return ALTRO_NOT_READY;
    }
  const auto t0 = std::chrono::steady_clock::now();
  altro_status first = ALTRO_OK;
  int launched = 0;
  for (; launched < ndev; ++launched) {
    const altro_status st = altro_solve_al_async(g->handle[launched]);  // one parked worker thread per handle
    if (st != ALTRO_OK) {
      first = st;
      g->err = std::string("altro_solve_al_async on part ") + std::to_string(launched) + ": " + altro_last_error(g->handle[launched]);
      break;
    }
  }
  // wait in completion order (poll), so that part_ms is each part's own wall time
  std::vector<char> done(ndev, 0);
  for (int left = launched; left > 0;) {
    for (int i = 0; i < launched; ++i) {
      if (done[i]) continue;
      int d = 0;
      altro_solve_poll(g->handle[i], &d);
      if (!d) continue;
      const altro_status st = altro_wait(g->handle[i]);
      g->part_ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (st != ALTRO_OK && first == ALTRO_OK) {
        first = st;
        g->err = std::string("solve on part ") + std::to_string(i) + ": " + altro_last_error(g->handle[i]);
      }
      done[i] = 1;
      --left;
    }
    // (the parts' worker threads and the device do the work; this thread only stamps completion times.  The latency path --
    //  a sub-millisecond solve of a few instances per device -- must not pay a 50 - 100 us timer sleep per solve: yield for
    //  the first 300 us, sleep only once the solve has proven long)
    if (left > 0) {
      if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 300.0)
        std::this_thread::yield();
      else
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  if (first != ALTRO_OK) return first;
  return altro_group_gather(g);
}

altro_status altro_group_get_results(altro_group g, int part, double* out, int capacity_records) {
  if (!g || !out || part < 0 || part >= (int)g->dev.size()) return ALTRO_INVALID_ARG;
  const int ndev = (int)g->dev.size();
  if (capacity_records < altro_group_total(g)) {
    g->err = "altro_group_get_results: the buffer holds fewer records than the global batch";
    return ALTRO_INVALID_ARG;
  }
  if (!g->slot || !g->d_recv[part]) {
    g->err = "nothing gathered yet (altro_group_solve_al / altro_group_gather)";
    return ALTRO_NOT_READY;
  }
  std::vector<double> all((size_t)ndev * g->slot * 4);
  GROUP_HIP(hipSetDevice(g->dev[part]));
  GROUP_HIP(hipMemcpyAsync(all.data(), g->d_recv[part], all.size() * sizeof(double), hipMemcpyDeviceToHost, g->stream[part]));
  GROUP_HIP(hipStreamSynchronize(g->stream[part]));
  size_t o = 0;
  for (int i = 0; i < ndev; ++i)  // trim the padding of the smaller parts
    for (int r = 0; r < g->batch[i]; ++r, ++o)
      for (int f = 0; f < 4; ++f) out[4 * o + f] = all[((size_t)i * g->slot + r) * 4 + f];
  return ALTRO_OK;
}

double altro_group_part_ms(altro_group g, int part) {
  return (g && part >= 0 && part < (int)g->part_ms.size()) ? g->part_ms[part] : 0.0;
}
double altro_group_gather_ms(altro_group g) { return g ? g->gather_ms : 0.0; }
const char* altro_group_last_error(altro_group g) { return g ? g->err.c_str() : g_create_err.c_str(); }

void altro_group_destroy(altro_group g) {
  if (!g) return;
  FreeBuffers(g);
  FreeTrajectoryBuffers(g);
  for (size_t i = 0; i < g->dev.size(); ++i) {
    if (g->comms_ok && g->comm[i]) ncclCommDestroy(g->comm[i]);
    if (g->stream[i] && hipSetDevice(g->dev[i]) == hipSuccess) hipStreamDestroy(g->stream[i]);
  }
  delete g;
}

}  // extern "C"
