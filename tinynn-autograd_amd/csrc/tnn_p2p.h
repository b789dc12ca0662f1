// Internal interface between tnn_comm.hip (the tnn_allreduce / tnn_allgather front) and the xGMI
// peer-to-peer transport in tnn_p2p.hip.
#pragma once
#include <stdint.h>

namespace tnn {
bool p2p_world(int* rank, int* world);                       // false when no peer group exists
bool p2p_can_allreduce(int64_t n, int dtype, int rop);       // enabled, f32 SUM, fits the mapped regions
int p2p_allreduce(float* buf, int64_t n);
bool p2p_can_allgather(int64_t n_per_rank, int dtype);       // enabled, <= 256 B per rank
int p2p_allgather(const void* send, void* recv, int64_t n_per_rank, int dtype);
}  // namespace tnn
