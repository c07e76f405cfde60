// Multi-GPU exchange of the search path: one process per GPU; the ranks of a job meet through a 128-byte id that the caller hands to
// every rank (environment, file, MPI, torch store -- any control plane), device buffers travel by RCCL over xGMI, small host values over the bootstrap sockets.
// A second transport moves device buffers through host memory and the same sockets, for ranks that share a GPU (tests on a one-GPU box: RCCL refuses two ranks
// on one device).  C ABI in include/spacepharer_amd.h (sp_dist_*, sp_ctx_allgather_*).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace sp {

enum { kTransportRccl = 0, kTransportHost = 1 };
constexpr size_t kDistIdBytes = 128;

class Comm {
public:
    ~Comm();
    // rank 0 listens on the id's port, the others connect (retrying for `timeoutSec`); with kTransportRccl the ranks then set up an RCCL communicator on the
    // current HIP device (the RCCL id travels over the sockets)
    static Comm *create(const unsigned char *id, int rank, int world, int transport, int timeoutSec, std::string &err);
    static bool uniqueId(unsigned char *id, std::string &err);
    int rank() const { return rank_; }
    int world() const { return world_; }
    int transport() const { return transport_; }
    // equal-size all-gather: recv holds world x bytes, rank r's block at r x bytes.  onDevice: both buffers are device memory of this rank's GPU
    bool allgather(const void *send, size_t bytes, void *recv, bool onDevice, std::string &err);
    // block k of `send` (sendBytes[k] bytes) goes to rank k, recv[k] takes rank k's block for this rank (recvBytes[k] bytes, agreed on beforehand)
    bool alltoallv(const void *const *send, const size_t *sendBytes, void *const *recv, const size_t *recvBytes, bool onDevice, std::string &err);
    bool barrier(std::string &err);
    bool allreduce(double *v, size_t n, int op /* 0 sum, 1 max, 2 min */, std::string &err);
    uint64_t bytesMoved() const { return bytes_; }

private:
    Comm() = default;
    bool hostAllgather(const void *send, size_t bytes, void *recv, std::string &err);
    int rank_ = 0, world_ = 1, transport_ = kTransportHost;
    int listenFd_ = -1;
    std::vector<int> fds_;            // rank 0: one per peer (index = rank); others: fds_[0] = the root
    void *nccl_ = nullptr; void *stream_ = nullptr;
    std::vector<unsigned char> stage_;
    uint64_t bytes_ = 0;
};

// DBReader::decomposeDomainByAminoAcid (lib/mmseqs/src/commons/DBReader.cpp:1171-1213): contiguous slices balanced by data size; lengths = the index `length` fields
// (sequence length + 2).  false: more ranks than data.
bool decomposeByResidues(const uint64_t *lengths, size_t n, int rank, int world, size_t &start, size_t &count);

}  // namespace sp
