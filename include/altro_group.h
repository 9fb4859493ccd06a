/* altro_group.h -- multi-GPU exchange of the batched solver for C / C++ callers.
 *
 * The reference (optimusride/altro-cpp) has no batch and no multi-device code: its perf driver solves one problem on
 * one thread (/root/reference/perf/benchmark_unicycle.cpp:18-43).  Instances of a batch are independent, so the batch
 * shards trivially: a contiguous block of instances per GPU, one altro_handle (include/altro_hip.h) and one host
 * thread per device, NO data-path collective, and after the solves ONE RCCL all-gather over xGMI of the 32-byte
 * per-instance result record {cost, violation, iterations_total, status} (4 doubles).  This library is that exchange
 * for a single process that drives several GPUs of one node (ncclCommInitAll): libaltro_group.so links libaltro_hip.so
 * and librccl.so; the solver library itself stays free of RCCL.  The multi-process form of the same exchange (one
 * rank per GPU, torch.distributed) is altro-cpp_amd/sharding.py, which bench.py uses.
 *
 * Ownership: the group owns its communicators, streams and gather buffers; the handles stay the caller's (attach a
 * handle created on the same device as the part; destroy the group before the handles).  Calls on one group must be
 * serialised by the caller.  Every function returns an altro_status; no exception crosses the ABI. */
#ifndef ALTRO_GROUP_H_
#define ALTRO_GROUP_H_

#include "altro_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct altro_group_s* altro_group;

/* Contiguous block split of `total` instances over `parts` devices: parts [0, total % parts) hold one more.  Pure
 * arithmetic (no device): the same split as sharding.shard_range and bench.py. */
void altro_group_shard_range(int total, int parts, int part, int* lo, int* hi);

/* One communicator, one stream and one record buffer per device of `device_ids` (ncclCommInitAll). */
altro_status altro_group_create(const int* device_ids, int ndev, altro_group* out);
int altro_group_size(altro_group g);
/* The handle that solves part `part` (its instances are the part's block of the global batch, `batch` of them);
 * it must have been created with device_id = device_ids[part]. */
altro_status altro_group_attach(altro_group g, int part, altro_handle h, int batch);
/* AugmentedLagrangianiLQR::Solve on every attached handle at once (altro_solve_al_async: one worker thread per
 * handle), then altro_group_gather.  Returns the first failure. */
altro_status altro_group_solve_al(altro_group g);
/* The exchange alone (after solves the caller ran itself): every handle packs its records on its own device
 * (altro_pack_results_device), ONE ncclAllGather on the group's streams, every device then holds the records of all
 * parts in part order. */
altro_status altro_group_gather(altro_group g);
/* The gathered records as the device of part `part` holds them: [sum of batches][4] doubles in part order. */
altro_status altro_group_get_results(altro_group g, int part, double* out, int capacity_records);
int altro_group_total(altro_group g);
/* The optional second collective of SURVEY.md section 8(e): whole trajectories on every device.  Every handle packs
 * X[b][N+1][n] and U[b][N][m] (doubles, the layout of altro_get_trajectory) on its own device
 * (altro_pack_trajectory_device), two ncclAllGather calls in one group on the group's streams.  n, m, N: the
 * dimensions the attached handles were created with.  C4: 32 768 x 503 doubles = 132 MB per device. */
altro_status altro_group_gather_trajectories(altro_group g, int n, int m, int N);
/* The gathered trajectories as the device of part `part` holds them, in part order: X[total][N+1][n], U[total][N][m]
 * (either may be NULL); capacity_instances >= altro_group_total. */
altro_status altro_group_get_trajectories(altro_group g, int part, double* X, double* U, int capacity_instances);
double altro_group_trajectory_gather_ms(altro_group g);
/* Wall time of the last altro_group_solve_al per part (ms; load imbalance is the only scaling loss) and of the
 * exchange. */
double altro_group_part_ms(altro_group g, int part);
double altro_group_gather_ms(altro_group g);
const char* altro_group_last_error(altro_group g);
void altro_group_destroy(altro_group g);

#ifdef __cplusplus
}
#endif
#endif  /* ALTRO_GROUP_H_ */
