/* graphite_mi355x_test.h — TEST / DIAGNOSTIC entry points of libgraphite_mi355x.so.
 *
 * Not part of the drop-in boundary (include/graphite_mi355x.h): nothing here replaces a reference
 * interface.  Used by tests/ and tools/ only. */
#ifndef GRAPHITE_MI355X_TEST_H
#define GRAPHITE_MI355X_TEST_H
#include "graphite_mi355x.h"
#ifdef __cplusplus
extern "C" {
#endif
/* in-process group of `n` landmark shards on ONE GPU, one host thread per shard afterwards: exercises the
 * sharded algorithm on a 1-GPU box (the product communicator is gr_bal_comm_init, RCCL) */
gr_status gr_bal_comm_init_local(gr_bal_problem **problems, int n);
/* all-reduce `n` doubles (host buffer, in place) through the problem's communicator: checks a transport against a
 * host sum (tests/test_gpu_ipc.py) */
gr_status gr_bal_comm_allreduce_host(gr_bal_problem *p, double *host_values, size_t n);
/* mean device time (us) of `reps` back-to-back launches of one hot kernel (tools/diag_*.py) */
double gr_bal_diag_time(gr_bal_problem *p, int which, int variant, int reps);
/* the DPP lane exchanges of csrc/common.hpp against __shfl_xor on every lane; *mismatches must come back 0 */
gr_status gr_test_lane_xor(int device, int *mismatches);
#ifdef __cplusplus
}
#endif
#endif
